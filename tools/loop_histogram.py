"""Opcode histogram of the step loop of one kernel variant (development aid: this is how the v_bitop3 / scalar-load
opportunities were found).

    python tools/loop_histogram.py even_rosenbrock "LaplaceProposalILi30EEELi30ELb1ELb0" [top_n]

Compiles rwm-pt-pytorch_amd/csrc/variants_<name>.hip with -save-temps, cuts the first kernel whose mangled name
contains the filter, and counts the instructions between the step-loop header and the swap section (both branches
of a two-way `scaled` dispatch are counted, so target-specific counts can be doubled)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, flt = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tmp = tempfile.mkdtemp(prefix="khist_")
src = os.path.join(ROOT, "rwm-pt-pytorch_amd", "csrc", f"variants_{name}.hip")
# the Makefile's flags for the max-ILP group (override with PTRWM_HIST_FLAGS)
flags = os.environ.get("PTRWM_HIST_FLAGS", "-mllvm -enable-post-misched=0 -mllvm -amdgpu-sched-strategy=max-ilp "
                                            "-fno-slp-vectorize").split()
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps=obj", "-c", src,
                       "-o", os.path.join(tmp, "v.o")] + flags, stderr=subprocess.DEVNULL, cwd=tmp)
lines = open(os.path.join(tmp, f"variants_{name}-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN5ptrwm17ptrwm_step_kernel") and flt in l and l.rstrip().endswith(
    ("EEvNS_5KArgsE:", "KArgsE")) or (l.startswith("_ZN5ptrwm17ptrwm_step_kernel") and flt in l and ":" in l))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
k = lines[start:end]
hdr = next(i for i, l in enumerate(k) if "This Loop Header: Depth=1" in l)
# the swap section starts with the barrier / wave-barrier pair after the accept block: take the first s_barrier
stop = next((i for i in range(hdr, len(k)) if "s_barrier" in k[i]), len(k))
ops = collections.Counter()
for l in k[hdr:stop]:
    m = re.match(r"\s+((?:v|s|ds|global|scratch|buffer|flat)_\w+)", l)
    if m:
        ops[m.group(1)] += 1
valu = sum(c for o, c in ops.items() if o.startswith("v_"))
print(f"{k[0][:100]}\nloop body lines {hdr}..{stop}: {sum(ops.values())} instructions, {valu} VALU")
for o, c in ops.most_common(top):
    print(f"{c:6d} {o}")
