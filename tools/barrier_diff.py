"""What the empty-asm value barriers do, statically (no GPU): the translation units of the surviving
kernels with the most spilled SGPRs are compiled twice from the same source - as shipped, and with
-DPTRWM_NO_VALUE_BARRIERS (philox.h: every barrier compiles to nothing) - and for each kernel of interest the code-object
metadata and the spill traffic in its text are put side by side.

    python tools/barrier_diff.py > profiles/r03_barrier_diff.txt

Read with profiles/r03_miscompile_trace.txt: the barriers are not on the def-use chain of the address that faulted in
round 1; this table shows what they buy (fewer live SGPRs across the step loop, i.e. fewer spill lanes and fewer
v_readlane_b32 inside the loop) and that removing them never lowers the register pressure of a kernel."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from issue_model import SCHED, kernel_blocks, step_path  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (translation unit, flags of its group, mangled-name filter, what it is)
CASES = [
    ("quad_full_rosenbrock", ["-fno-slp-vectorize"], "quad_step_kernelINS_15QFullRosenbrockILi28ELin1EEENS_8QLaplaceILi28ELin1EEELi28ELi0ELi512ELb0ELb0EEE",
     "lane-split FullRosenbrock W=28 run-time dim + Laplace, production (most spilled SGPRs of all float-state production kernels)"),
    ("quad_hybrid_rosenbrock", ["-fno-slp-vectorize"], "quad_step_kernelINS_17QHybridRosenbrockILi28ELin1EEENS_8QLaplaceILi28ELin1EEELi28ELi0ELi512ELb0ELb0EEE",
     "lane-split HybridRosenbrock W=28 run-time dim + Laplace, production"),
    ("three_mixture", SCHED, "ptrwm_step_kernelINS_12ThreeMixtureILi64EEENS_21UniformRadiusProposalILi64EEELi64ELb0ELb1EEE",
     "thread form ThreeMixture width 64 + UniformRadius, FIXTURE variant (most spilled SGPRs of the thread form)"),
    ("three_mixture1", SCHED, "ptrwm_step_kernelINS_13ThreeMixture1ILi50EEENS_21UniformRadiusProposalILi50EEELi50ELb1ELb0ELb0EEE",
     "thread form ThreeMixture1 dim 50 + UniformRadius, production (BASELINE configs[4])"),
    ("rough_carpet2", SCHED, "ptrwm_step_kernelINS_12RoughCarpetTILi30ELb1EEENS_14NormalProposalILi30EEELi30ELb1ELb0ELb0EEE",
     "thread form RoughCarpet2 dim 30 + Normal, production (BASELINE configs[2], the headline)"),
]


def build(tu, flags, extra):
    tmp = tempfile.mkdtemp(prefix="barrier_")
    src = os.path.join(ROOT, "rwm-pt-pytorch_amd", "csrc", ("" if tu.startswith("quad_") else "variants_") + tu + ".hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                           "-save-temps=obj", "-c", src, "-o", os.path.join(tmp, "v.o")] + flags + extra, stderr=subprocess.DEVNULL, cwd=tmp)
    return os.path.join(tmp, os.path.basename(src).replace(".hip", "-hip-amdgcn-amd-amdhsa-gfx950.s"))


def facts(asm, flt):
    txt = open(asm).read()
    name, blocks, order = kernel_blocks(asm, flt)
    meta = {}
    for rec in re.split(r"\n  - \.agpr_count", txt)[1:]:
        m = re.search(r"\.name:\s+(\S+)", rec)
        if m and m.group(1) == name:
            for k in ("vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size"):
                meta[k] = int(re.search(r"\." + k + r":\s+(\d+)", rec).group(1))
            meta["agpr_count"] = int(re.match(r":\s+(\d+)", rec).group(1))
    ops = [o for b in order for o, _ in blocks[b]["ins"]]
    header, path, _ = step_path(blocks, order)
    loop_ops = [o for b in path for o, _ in blocks[b]["ins"]]
    return {**meta, "instructions": len(ops), "v_writelane": ops.count("v_writelane_b32"), "v_readlane": ops.count("v_readlane_b32"),
            "step_path_valu": sum(o.startswith("v_") for o in loop_ops), "step_path_v_readlane": loop_ops.count("v_readlane_b32"),
            "step_path_s_load": sum(o.startswith("s_load") for o in loop_ops)}


def main():
    print(__doc__)
    jobs = {}
    with ThreadPoolExecutor(max_workers=4) as ex:
        for tu, flags, _, _ in CASES:
            for tag, extra in (("shipped", []), ("no barriers", ["-DPTRWM_NO_VALUE_BARRIERS"])):
                if (tu, tag) not in jobs:
                    jobs[(tu, tag)] = ex.submit(build, tu, flags, extra)
    keys = ("vgpr_count", "agpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size", "instructions",
            "v_writelane", "v_readlane", "step_path_valu", "step_path_v_readlane", "step_path_s_load")
    for tu, flags, flt, what in CASES:
        a, b = facts(jobs[(tu, "shipped")].result(), flt), facts(jobs[(tu, "no barriers")].result(), flt)
        print(f"\n{what}\n  {flt}")
        print(f"  {'':28s} {'shipped':>10s} {'no barriers':>12s}")
        for k in keys:
            print(f"  {k:28s} {a.get(k, 0):10d} {b.get(k, 0):12d}")
    print("\nstep_path_* = the instructions one Metropolis step without a swap event executes (tools/issue_model.py).")


if __name__ == "__main__":
    main()
