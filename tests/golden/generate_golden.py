#!/usr/bin/env python3
"""Generates the golden ExaSlang vectors under tests/golden/ by running the REFERENCE's own Python
lowering in this container. Run from the repo root:  python tests/golden/generate_golden.py

What is real and what is stubbed
  * real, imported unmodified from /root/reference: evostencils.ir.*, evostencils.stencils.*,
    evostencils.grammar.multigrid (OperatorInfo/EquationInfo, generate_primitive_set, add_level and
    its production closures) and evostencils.code_generation.exastencils.ProgramGenerator
    (generate_storage, generate_cycle_function, generate_multigrid).
  * stubbed in-process (nothing is written outside this repo): the `deap` package, which is absent
    here and which the lowering never calls; only `deap.gp.PrimitiveSetTyped` is needed as a base
    class (grammar/gp.py:3,55). The stub records addPrimitive/addTerminal registrations so the
    reference's production closures can be called by name, as DEAP's gp.compile would.
  * bypassed: ProgramGenerator.__init__ (it starts the ExaStencils JVM, exastencils.py:93); the
    instance is created with __new__ and given exactly the attributes the lowering reads.
  * the problem information (equations/operators/fields), which the reference parses from
    ExaStencils' debug output (parser.py:25-111), is constructed from the problem files' formulas.

Also extracts the evolved cycle recorded in notebooks/tutorial.ipynb (cell output, lines 3421-3470)
— a recorded OUTPUT of the reference, i.e. data — into tutorial_recorded_cycle.exa3.

The outputs (strings) are committed; the reference itself never ships.
"""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def install_deap_stub():
    deap = types.ModuleType("deap")
    gp = types.ModuleType("deap.gp")

    class PrimitiveSetTyped:
        def __init__(self, name, in_types, ret_type, prefix="ARG"):
            self.name, self.ins, self.ret = name, in_types, ret_type
            self.context = {}
            # typed registrations (for tests/test_reference_grammar_fuzz.py): return type -> what produces it
            self.typed_primitives = {}       # ret_type -> [(name, callable, [argument types])]
            self.typed_terminals = {}        # ret_type -> [(name or None, object)]

        def addPrimitive(self, primitive, in_types, ret_type, name=None):
            self.context[name or primitive.__name__] = primitive
            self.typed_primitives.setdefault(ret_type, []).append((name or primitive.__name__, primitive, list(in_types)))

        def addTerminal(self, terminal, ret_type, name=None):
            if name is not None:
                self.context[name] = terminal
            self.typed_terminals.setdefault(ret_type, []).append((name, terminal))

    class Primitive:
        pass

    gp.PrimitiveSetTyped, gp.Primitive = PrimitiveSetTyped, Primitive
    deap.gp = gp
    sys.modules["deap"], sys.modules["deap.gp"] = deap, gp


def reference_context(problem, min_level, max_level, return_pset=False):
    """(context dict of the reference's productions — or the primitive set itself —, ProgramGenerator instance, storages)."""
    import sympy
    from evostencils.ir import base, system
    from evostencils.stencils import constant
    from evostencils.grammar import multigrid as mg
    from evostencils.code_generation.exastencils import ProgramGenerator

    sys.path.insert(0, REPO)
    from evostencils_amd import problems as our_problems

    pd = our_problems.BY_NAME[problem](min_level, max_level)
    d = pd.dimension
    equations, operators = [], []
    fw = our_problems._fw_restriction_stencil(d)
    pr = our_problems._linear_prolongation_stencil(d)
    if pd.system is not None:
        # a system of PDEs: the equation texts and named stencils of the problem description, handed to the REFERENCE's
        # EquationInfo / OperatorInfo (what its parser extracts from ExaStencils' debug layer 3, parser.py:25-111)
        spec = pd.system
        fields = [sympy.Symbol(f) for f in spec.field_names]
        for level in range(min_level, max_level + 1):
            for i, text in enumerate(spec.equation_text(level)):
                eq = mg.EquationInfo(spec.equation_names[i], level, text)
                eq._associated_field = fields[i]
                equations.append(eq)
            for name, entries in spec.operator_stencils(level).items():
                operators.append(mg.OperatorInfo(name, level, constant.Stencil(entries, d)))
            for i, f in enumerate(fields):
                r = mg.OperatorInfo(spec.restriction_names[i], level, constant.Stencil(fw.entries, d), base.Restriction)
                r._associated_field = f
                operators.append(r)
                p_ = mg.OperatorInfo(spec.prolongation_names[i], level, constant.Stencil(pr.entries, d), base.Prolongation)
                p_._associated_field = f
                operators.append(p_)
    else:
        fields = [sympy.Symbol(pd.field_name)]
        offs = our_problems._star_offsets(d)
        for level in range(min_level, max_level + 1):
            eq = mg.EquationInfo(pd.equation_name, level,
                                 f"( {pd.operator_name}@{level} * {pd.field_name}@{level} ) == {pd.rhs_name}@{level}")
            eq._associated_field = fields[0]
            equations.append(eq)
            coeffs = pd.mg_stencil(level, pd.parameters)[:1 + 2 * d]
            operators.append(mg.OperatorInfo(pd.operator_name, level,
                                             constant.Stencil(list(zip(offs, [float(c.real) if isinstance(c, complex) else float(c) for c in coeffs])), d)))
            # ExaStencils declares the default inter-grid operators on every level; the reference asserts one
            # restriction and one prolongation per field and level (grammar/multigrid.py:156-157)
            operators.append(mg.OperatorInfo(pd.restriction_name, level, constant.Stencil(fw.entries, d), base.Restriction))
            operators.append(mg.OperatorInfo(pd.prolongation_name, level, constant.Stencil(pr.entries, d), base.Prolongation))

    g = ProgramGenerator.__new__(ProgramGenerator)
    g._cycle_name = "gen_mgCycle"
    g._solver_cache = {}
    g._use_jacobi_prefix = True
    g._dimension = d
    g._equations, g._fields, g._operators = equations, fields, operators
    g._coarsening_factor = [(2,) * d for _ in fields]
    g._min_level, g._max_level = min_level, max_level
    finest = [base.Grid((2 ** max_level,) * d, (1 / 2 ** max_level,) * d, max_level) for _ in fields]
    storages = g.generate_storage(min_level, max_level, finest)

    approximation = system.Approximation("x", [base.Approximation(f.name, gr) for f, gr in zip(fields, finest)])
    top = [eq for eq in equations if eq.level == max_level]
    rhs = system.RightHandSide("b", [base.RightHandSide(eq.rhs_name, gr) for eq, gr in zip(top, finest)])
    pset, _ = mg.generate_primitive_set(approximation, rhs, d, g._coarsening_factor, max_level, equations,
                                        operators, fields, maximum_local_system_size=8,
                                        depth=max_level - min_level)
    return (pset if return_pset else pset.context), g, storages


def fas_reference_context(depth, return_pset=False):
    """(context of the reference's FAS productions, the reference's ProgramGeneratorFAS) for this repository's own FAS
    problem files (tests/golden/problems/FAS_2D_Exponential). The reference class is constructed normally: its constructor
    only reads the .knowledge file and the template (no JVM, exastencils_FAS.py:12-96)."""
    from evostencils.ir import base, system
    from evostencils.grammar import multigrid as mg
    from evostencils.code_generation.exastencils_FAS import ProgramGeneratorFAS
    g = ProgramGeneratorFAS("FAS_2D_Exponential", "Solution", "RHS", "Residual", "Approximation", "RestrictionNode",
                            "CorrectionNode", "Laplace", "gamSten", "gen_mgCycle", "CGS",
                            build_path=os.path.join(HERE, "problems"))
    approximation = system.Approximation("x", [base.Approximation(f.name, gr) for f, gr in zip(g.fields, g.finest_grid)])
    top = [eq for eq in g.equations if eq.level == g.max_level]
    rhs = system.RightHandSide("b", [base.RightHandSide(eq.rhs_name, gr) for eq, gr in zip(top, g.finest_grid)])
    pset, _ = mg.generate_primitive_set(approximation, rhs, g.dimension, g.coarsening_factor, g.max_level, g.equations,
                                        g.operators, g.fields, maximum_local_system_size=8, depth=depth, FAS=True)
    return (pset if return_pset else pset.context), g


def extract_tutorial_cycle():
    with open(os.path.join(REFERENCE, "notebooks", "tutorial.ipynb")) as fh:
        nb = json.load(fh)
    for cell in nb["cells"]:
        for out in cell.get("outputs", []):
            text = "".join(out.get("text", []))
            if "ExaSlang Code:" in text and "Function gen_mgCycle@9" in text:
                start = text.index("Function gen_mgCycle@9")
                end = text.index("\n}\n", start) + 3
                return text[start:end]
    raise RuntimeError("recorded cycle not found in tutorial.ipynb")


def main():
    sys.dont_write_bytecode = True
    install_deap_stub()
    sys.path.insert(0, REFERENCE)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import cycles

    manifest = {}
    for name, (problem, min_level, max_level, builder) in cycles.CASES.items():
        ctx, gen, storages = reference_context(problem, min_level, max_level)
        expression, _ = builder(ctx)
        text = gen.generate_cycle_function(expression, storages, min_level, max_level, max_level)
        # lowering twice must give the same text (valid flags are restored, exastencils.py:330-334)
        again = gen.generate_cycle_function(expression, storages, min_level, max_level, max_level)
        assert text == again
        path = os.path.join(HERE, f"{name}.exa3")
        with open(path, "w") as fh:
            fh.write(text)
        manifest[name] = {"problem": problem, "min_level": min_level, "max_level": max_level,
                          "file": f"{name}.exa3", "lines": text.count("\n")}
        print(f"{name}: {text.count(chr(10))} lines")
    # non-linear FAS backend (exastencils_FAS.py): layer-4 text of grammar-built cycles
    for name, (depth, builder) in cycles.FAS_CASES.items():
        ctx, gen = fas_reference_context(depth)
        expression, _ = builder(ctx)
        text = gen.generate_cycle_function(expression)
        with open(os.path.join(HERE, f"{name}.exa4"), "w") as fh:
            fh.write(text)
        manifest[name] = {"problem": "FAS_2D_Exponential (tests/golden/problems)", "depth": depth, "file": f"{name}.exa4",
                          "lines": text.count("\n")}
        print(f"{name}: {text.count(chr(10))} lines")
    # a RECORDED output of the reference's FAS generator (an earlier version of it): the cycle function appended to
    # example_problems/FAS_2D_Basic/FAS_2D_Basic.exa4 behind the marker "//generated by EvoStencils" — data, like the
    # tutorial's recorded cycle
    with open(os.path.join(REFERENCE, "example_problems", "FAS_2D_Basic", "FAS_2D_Basic.exa4")) as fh:
        text = fh.read()
    recorded_fas = text[text.index("Function gen_mgCycle@5", text.index("//generated by EvoStencils")):].rstrip() + "\n"
    with open(os.path.join(HERE, "fas_recorded_cycle.exa4"), "w") as fh:
        fh.write(recorded_fas)
    manifest["fas_recorded_cycle"] = {"source": "example_problems/FAS_2D_Basic/FAS_2D_Basic.exa4:213-270 (generated cycle recorded in the file)",
                                      "file": "fas_recorded_cycle.exa4"}
    recorded = extract_tutorial_cycle()
    with open(os.path.join(HERE, "tutorial_recorded_cycle.exa3"), "w") as fh:
        fh.write(recorded)
    manifest["tutorial_recorded_cycle"] = {"source": "notebooks/tutorial.ipynb:3421-3470 (recorded cell output)",
                                           "file": "tutorial_recorded_cycle.exa3"}
    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=2, sort_keys=True)


if __name__ == "__main__":
    main()
