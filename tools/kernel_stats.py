"""Register / spill / LDS summary of the step-kernel variants in one translation unit (development aid).

    python tools/kernel_stats.py rough_carpet2 [filter-substring]

Compiles rwm-pt-pytorch_amd/csrc/variants_<name>.hip with -save-temps and parses the code-object metadata."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "rough_carpet2"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp(prefix="kstats_")
src = os.path.join(ROOT, "rwm-pt-pytorch_amd", "csrc", f"variants_{name}.hip")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps=obj", "-c", src,
                       "-o", os.path.join(tmp, "v.o")] + sys.argv[3:], stderr=subprocess.DEVNULL)
txt = open(os.path.join(tmp, f"variants_{name}-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
for rec in re.split(r"\n  - \.agpr_count", txt)[1:]:
    m = re.search(r"\.name:\s+(\S+)", rec)
    if not m or "step_kernel" not in m.group(1) or flt not in m.group(1):
        continue
    n = m.group(1)
    k = re.search(r"(\w+Proposal)ILi(\d+)EEELi\d+ELb([01])ELb([01])", n)
    g = lambda key: re.search(key + r":\s+(\d+)", rec).group(1)  # noqa: E731
    print(f"{k.group(1):22s} D={k.group(2):>3s} exact={k.group(3)} full={k.group(4)} vgpr={g(r'.vgpr_count'):>3s} "
          f"vspill={g(r'.vgpr_spill_count'):>3s} sgpr={g(r'.sgpr_count'):>3s} sspill={g(r'.sgpr_spill_count'):>3s} "
          f"lds={g(r'.group_segment_fixed_size')}")
