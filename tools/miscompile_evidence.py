"""Regenerates, WITHOUT a GPU, the static evidence for the round-1 miscompile of the width-80 fixture kernel
(ptrwm_step_kernel<RoughCarpetT<80,true>, UniformRadiusProposal<80>, 80, false, true>; SIGBUS in
gpurun_out/gdb.log of round 1, commit 28bda15): the wide width group of variants_rough_carpet2.hip is compiled twice
from the same source - with the Makefile's max-ILP flags (the build that faulted) and with the default scheduler (the
build that ships) - and for both the code-object metadata of the kernel and the instruction window around the
`ext_u[srep]` load (the faulting access: gdb showed s[6:7] = the valid ext_u base, so the wild part of the address is
the VGPR pair holding srep) are written to profiles/.  Nothing is executed; the fault is not re-run.

    python tools/miscompile_evidence.py > profiles/r02_miscompile_width80.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rwm-pt-pytorch_amd", "csrc", "variants_rough_carpet2.hip")
KERNEL = ("_ZN5ptrwm17ptrwm_step_kernelINS_12RoughCarpetTILi80ELb1EEENS_21UniformRadiusProposalILi80EEELi80ELb0ELb1EEEvNS_5KArgsE")
BASE = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DPTRWM_PART_WIDE", "-save-temps=obj", "-c", SRC]
MAXILP = ["-mllvm", "-enable-post-misched=0", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fno-slp-vectorize"]
GDB = """rocgdb, round 1 (gpurun_out/gdb.log, precise memory faults on):
  Thread 68 "ptrwm_step_-9a99" received signal SIGBUS, Bus error.
  in ptrwm::ptrwm_step_kernel<ptrwm::RoughCarpetT<80, true>, ptrwm::UniformRadiusProposal<80>, 80, false, true>(ptrwm::KArgs)
  host-side pointers: ext_prop 7ffe14800600 ext_u 7ffe14802c00 ext_swap_u 0 state 7ffe14800000 logp 7ffe14800400
  s6 = 0x14802c00, s7 = 0x7ffe   -> s[6:7] == ext_u (valid)
  address VGPR pair printed by gdb: 0x188cc784:0x10326e1 ... (not inside any allocation: srep is garbage)"""


def build(flags):
    tmp = tempfile.mkdtemp(prefix="mc_")
    subprocess.check_call(BASE + flags + ["-o", os.path.join(tmp, "v.o")], stderr=subprocess.DEVNULL, cwd=tmp)
    return open(os.path.join(tmp, "variants_rough_carpet2-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def report(tag, txt):
    print(f"==== {tag}")
    for rec in re.split(r"\n  - \.agpr_count", txt)[1:]:
        m = re.search(r"\.name:\s+(\S+)", rec)
        if not m or m.group(1) != KERNEL:
            continue
        g = lambda key: re.search(key + r":\s+(\d+)", rec).group(1)  # noqa: E731
        agpr = re.match(r":\s+(\d+)", rec).group(1)
        vals = {k: g("." + k) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count",
                                        "private_segment_fixed_size")}
        print(f"metadata: agpr_count {agpr}  " + "  ".join(f"{k} {v}" for k, v in vals.items()))
    i = txt.index(KERNEL + ":")
    body = txt[i:txt.index(".end_amdhsa_kernel", i)].splitlines()
    ops = {}
    for ln in body:
        m = re.match(r"\s+(\w+)", ln)
        if m:
            ops[m.group(1)] = ops.get(m.group(1), 0) + 1
    print("SGPR spill traffic: v_writelane_b32", ops.get("v_writelane_b32", 0), " v_readlane_b32", ops.get("v_readlane_b32", 0),
          " v_accvgpr_write_b32", ops.get("v_accvgpr_write_b32", 0), " v_accvgpr_read_b32", ops.get("v_accvgpr_read_b32", 0))
    # the step loop header: the first `v_lshl_add_u64 ..., 2, s[N:N+1]` whose result feeds the next global_load_dword is
    # ext_u + 4 * srep (kernel.h: `ext_u = a.full.ext_u[srep]`)
    for k, ln in enumerate(body):
        if re.search(r"v_lshl_add_u64 v\[\d+:\d+\], v\[\d+:\d+\], 2, s\[\d+:\d+\]", ln) and k + 1 < len(body) \
                and "global_load_dword" in body[k + 1] and "Loop Header" in "\n".join(body[max(0, k - 60):k]):
            print(f"instruction window around the ext_u[srep] load (kernel line {k}):")
            for j in range(max(0, k - 42), k + 3):
                print(f"  {j:6d} {body[j]}")
            break
    print()


if __name__ == "__main__":
    print(__doc__)
    print(GDB)
    print()
    report("max-ILP build (-mllvm -enable-post-misched=0 -mllvm -amdgpu-sched-strategy=max-ilp -fno-slp-vectorize): FAULTED", build(MAXILP))
    report("default-scheduler build (what csrc/Makefile uses for PTRWM_PART_WIDE): passes all variants + fuzz", build([]))
    print("Reading: both builds need > 256 VGPRs (AGPRs in use) and spill ~210 SGPRs into lanes of v252/v253; the address\n"
          "pair of the faulting load is rebuilt every step from SGPRs reloaded with v_readlane_b32 out of those lanes.\n"
          "The two builds differ ONLY in the scheduler flags; the source (including the `asm volatile(\"\" : \"+s\")`\n"
          "value barriers in kernel.h / philox.h) is identical, and the default-scheduler build of the same kernel passes\n"
          "tools/check_all_variants.py and 3 800 fuzz cases.  That makes the flag combination the trigger; it does not\n"
          "prove the barriers innocent of contributing, so tools/kernel_stats.py --check now refuses ANY kernel in the\n"
          "max-ILP group with agpr_count > 0 or vgpr_count > 256 (the regime in which it happened).")
