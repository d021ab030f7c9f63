"""Developer tool: instruction mix of the innermost loops of one kernel in a disassembly (llvm-objdump -d of the gfx950 code object).
usage: python tools/count_loop_isa.py <dev.s> <mangled-name-substring>
A loop = a backward branch (s_cbranch_* to a lower address); for each, counts by class between target and branch."""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*%s" % re.escape(sys.argv[2]), l))
end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
body = []
for l in lines[start + 1:end]:
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
    if m:
        body.append((int(m.group(3), 16), m.group(1), m.group(2)))
addr_index = {a: i for i, (a, _, _) in enumerate(body)}


def klass(op):
    if op.endswith("_f64") or "_f64_" in op:
        return "valu_f64"
    if "dpp" in op:
        return "valu_dpp"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store")):
        return "vmem_store"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    return "other"


for i, (a, op, args) in enumerate(body):
    if op.startswith("s_cbranch") or op == "s_branch":
        tgt = None
        m2 = re.match(r"(\d+)", args.split()[0]) if args else None
        if m2:
            off = int(m2.group(1))
            if off >= 1 << 15:
                off -= 1 << 16
            tgt = a + 4 + 4 * off
        if tgt is not None and tgt < a and tgt in addr_index:
            j = addr_index[tgt]
            c = Counter(klass(o) for _, o, s in body[j:i + 1])
            dpp = sum(1 for _, o, s in body[j:i + 1] if "dpp" in s or "dpp" in o)
            n = i + 1 - j
            if n > 40:
                print("loop of %d instructions at +0x%x: %s; with a dpp modifier: %d" % (n, tgt - body[0][0], dict(c), dpp))
