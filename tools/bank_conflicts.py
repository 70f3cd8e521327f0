"""Static VGPR bank-conflict score of a kernel's step loop (development aid).

    python tools/bank_conflicts.py file.s [more.s ...]

For every VALU instruction between the step-loop header and the swap section: source VGPRs are mapped to banks
(register index mod 4); an instruction whose distinct source registers share a bank scores one per extra register in
that bank.  The same instruction sequence with different register assignments measured 5.20 vs 5.67 ms."""
import re
import sys


def score(path):
    lines = open(path).read().split("\n")
    a = next(i for i, l in enumerate(lines) if "This Loop Header: Depth=1" in l)
    b = next((i for i in range(a, len(lines)) if "logpILb0" in lines[i] and "exit" in lines[i]), len(lines))
    tot = n = 0
    by_op = {}
    for l in lines[a:b]:
        m = re.match(r"\s+(v_\w+)\s+(.*)", l)
        if not m:
            continue
        op, rest = m.group(1), m.group(2).split(";")[0]
        ops = [o.strip() for o in rest.split(",")]
        srcs = ops[1:]  # first operand is the destination
        if op.startswith("v_cmp") and not op.endswith("_e64"):
            srcs = ops  # e32 compares write vcc implicitly
        if op.startswith("v_mad_u64_u32"):
            srcs = ops[2:]  # vdst, sdst, then sources
        regs = set()
        for o in srcs:
            mm = re.match(r"v\[(\d+):(\d+)\]", o)
            if mm:
                regs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
            else:
                mm = re.match(r"v(\d+)$", o)
                if mm:
                    regs.add(int(mm.group(1)))
        banks = {}
        for r in regs:
            banks[r % 4] = banks.get(r % 4, 0) + 1
        c = sum(v - 1 for v in banks.values() if v > 1)
        tot += c
        n += 1
        if c:
            by_op[op] = by_op.get(op, 0) + c
    return tot, n, by_op


for p in sys.argv[1:]:
    t, n, by = score(p)
    print(f"{p}: conflict score {t} over {n} VALU instructions; by opcode: "
          + ", ".join(f"{k} {v}" for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]))
