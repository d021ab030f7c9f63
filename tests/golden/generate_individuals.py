#!/usr/bin/env python3
"""Generates tests/golden/grammar_individuals.json: random individuals of the reference's grammar as STRINGS (the form DEAP
prints and the optimizer stores), grown over the reference's own typed primitive sets with the generator of
tests/test_reference_grammar_fuzz.py and kept only if the reference's own lowering accepts them. Data for the GPU parity
tests (the reference is absent there): the strings are compiled with this repository's productions and solved on the GPU
and by the oracle. Run from the repo root in the build container:  python tests/golden/generate_individuals.py"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE), HERE]
import generate_golden as gg                                                    # noqa: E402

SETS = {"2D_FD_Poisson_fromL2": (3, 6, 24), "3D_FD_Poisson_fromL2": (2, 5, 16), "2D_FD_LinearElasticity_fromL2": (3, 6, 24)}


def main():
    sys.dont_write_bytecode = True
    gg.install_deap_stub()
    sys.path.insert(0, gg.REFERENCE)
    from test_reference_grammar_fuzz import Grower
    out = {}
    for problem, (lo, hi, count) in SETS.items():
        rng = random.Random(31337 + hi)
        strings = []
        while len(strings) < count:
            pset, ref_gen, storages = gg.reference_context(problem, lo, hi, return_pset=True)
            try:
                state, text = Grower(pset, rng).grow_with_string(pset.ret, budget=rng.randint(7, 15))
                ref_gen.generate_cycle_function(state[0], storages, lo, hi, hi)
            except Exception:                                                    # noqa: BLE001
                continue
            if text not in strings:
                strings.append(text)
        out[problem] = {"depth": hi - lo, "individuals": strings}
        print(problem, len(strings), "individuals,", sum(len(s) for s in strings) // len(strings), "characters on average")
    with open(os.path.join(HERE, "grammar_individuals.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
