"""Opcode histogram of the step loop of one lane-split kernel variant (development aid).

    python tools/quad_histogram.py rough_carpet2 "QNormalILi8ELi6EEELi8ELi30ELb0" [top_n]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, flt = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tmp = tempfile.mkdtemp(prefix="qhist_")
src = os.path.join(ROOT, "rwm-pt-pytorch_amd", "csrc", f"quad_{name}.hip")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-save-temps=obj",
                       "-c", src, "-o", os.path.join(tmp, "q.o")] + sys.argv[4:], stderr=subprocess.DEVNULL, cwd=tmp)
txt = open(os.path.join(tmp, f"quad_{name}-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
kname = next(l.split(":")[0] for l in txt.split("\n") if l.startswith("_ZN5ptrwm22ptrwm_quad_step_kernel") and ":" in l and flt in l)
i = txt.index(kname + ":")
body = txt[i:txt.index(".end_amdhsa_kernel", i)].split("\n")
hdr = next(k for k, l in enumerate(body) if "=>This Loop Header: Depth=1" in l)
lab = body[hdr].split(":")[0].strip()
ends = [k for k, l in enumerate(body) if re.search(r"s_c?branch\w* " + re.escape(lab) + r"\b", l)]
seg = body[hdr:ends[-1] + 1]
ops = collections.Counter()
for l in seg:
    m = re.match(r"\s+((?:v|s|ds|global|scratch|buffer|flat)_\w+)", l)
    if m:
        ops[m.group(1)] += 1
valu = sum(c for o, c in ops.items() if o.startswith("v_"))
print(f"{kname}\nstep loop: {sum(ops.values())} instructions, {valu} VALU (static count, all branches)")
for o, c in ops.most_common(top):
    print(f"{c:6d} {o}")
out = os.path.join(tmp, "loop.s")
open(out, "w").write("\n".join(seg))
print("loop body written to", out)
