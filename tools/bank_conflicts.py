"""Static VGPR bank-conflict score of a kernel's step loop (development aid).

    python tools/bank_conflicts.py file.s [more.s ...]

The same instruction sequence with different register assignments measured 5.20 vs 5.67 ms per launch."""
import re
import sys


def vregs(operand):
    mm = re.match(r"v\[(\d+):(\d+)\]", operand)
    if mm:
        return list(range(int(mm.group(1)), int(mm.group(2)) + 1))
    mm = re.match(r"v(\d+)$", operand)
    return [int(mm.group(1))] if mm else []


def score(path):
    """Measured on gfx950 (tools/bank_ubench.hip): a VALU instruction that reads three VGPR operands costs 4.4 instead
    of 2.45 cycles when two of them sit in the same bank (index mod 4) - or are the same register; two-operand
    instructions do not care.  Score = number of three-operand instructions with such a clash."""
    lines = open(path).read().split("\n")
    a = next(i for i, l in enumerate(lines) if "This Loop Header: Depth=1" in l)
    b = next((i for i in range(a, len(lines)) if "logpILb0" in lines[i] and "exit" in lines[i]), len(lines))
    tot = n3 = n = 0
    by_op = {}
    for l in lines[a:b + 220]:
        m = re.match(r"\s+(v_\w+)\s+(.*)", l)
        if not m:
            continue
        op, rest = m.group(1), m.group(2).split(";")[0]
        if op.startswith("v_pk_"):
            continue  # two passes anyway
        ops = [o.strip() for o in rest.split(",")]
        srcs = ops[1:]  # first operand is the destination
        if op.startswith("v_cmp") and not op.endswith("_e64"):
            srcs = ops
        if op.startswith("v_mad_u64_u32"):
            srcs = ops[2:]  # vdst, sdst, then sources
        if op.startswith(("v_fmac", "v_mac")):
            srcs = ops[1:] + [ops[0]]  # the destination is the addend
        regs = []
        for o in srcs:
            r = vregs(o)
            regs.extend(r[:1] if op.startswith("v_mad_u64_u32") and len(r) == 2 else r)
        n += 1
        if len(regs) < 3:
            continue
        n3 += 1
        banks = [r % 4 for r in regs]
        if len(set(banks)) < len(banks):
            tot += 1
            by_op[op] = by_op.get(op, 0) + 1
    return tot, n3, n, by_op


for p in sys.argv[1:]:
    t, n3, n, by = score(p)
    print(f"{p}: {t} of {n3} three-operand instructions clash ({n} VALU in the step loop); by opcode: "
          + ", ".join(f"{k} {v}" for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]))
