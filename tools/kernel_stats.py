"""Register / spill / LDS figures of the compiled kernels, read from the code-object metadata of the BUILT objects
(rwm-pt-pytorch_amd/build/*.o: .hip_fatbin -> gfx950 code object -> NT_AMDGPU_METADATA), no recompilation.

    python tools/kernel_stats.py [--objdir DIR] [filter-substring]     table of every kernel whose name contains the filter
    python tools/kernel_stats.py --check [--objdir DIR]                the build gate run by csrc/Makefile

--check fails (exit 1) if any kernel of the MAX-ILP group (objects built with $(SCHED): every variants_*.o that is not
*.wide.o) needs more than 256 VGPRs or any AGPR.  That is the regime in which hipcc's max-ILP scheduling miscompiled
the width-80 fixture kernel in round 1 (profiles/r02_miscompile_width80.txt): a new register width, or a kernel that
grows, must move to the default-scheduler group (PTRWM_WIDTHS_WIDE in variants.h) instead of silently landing there.
It also fails if a PRODUCTION step kernel (FULL = false) of that group uses scratch memory (DESIGN.md 3.1 promises
none; round 2 found and fixed 16 + 4 DP bytes per thread in every HybridRosenbrock kernel this way); scratch in the
default-scheduler group (widths > 64, one wave per SIMD) is reported, not fatal."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("agpr_count", "vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size")


def kernels_of(obj):
    """[(name, {field: int})] for every kernel in the gfx950 code object embedded in a host object file."""
    with tempfile.TemporaryDirectory(prefix="kstats_") as tmp:
        fat, co = os.path.join(tmp, "fat"), os.path.join(tmp, "co")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True)
        if r.returncode != 0 or not os.path.exists(fat):
            return []
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for rec in re.split(r"\n  - \.agpr_count:", notes)[1:]:
        rec = ".agpr_count:" + rec
        name = re.search(r"\.name:\s+(\S+)", rec).group(1)
        out.append((name, {f: int(re.search(r"\." + f + r":\s+(\d+)", rec).group(1)) for f in FIELDS}))
    return out


def short(name):
    r = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    r = r.replace("ptrwm::", "").replace("void ", "")
    return re.sub(r"\(KArgs\)$|\(.*\)$", "", r)


def main():
    args = sys.argv[1:]
    check = "--check" in args
    args = [a for a in args if a != "--check"]
    objdir = os.path.join(ROOT, "rwm-pt-pytorch_amd", "build")
    if "--objdir" in args:
        i = args.index("--objdir")
        objdir = args[i + 1]
        del args[i:i + 2]
    flt = args[0] if args else ""
    objs = sorted(glob.glob(os.path.join(objdir, "*.o")))
    if not objs:
        sys.exit(f"no objects under {objdir}: build first (make -C rwm-pt-pytorch_amd/csrc)")
    bad, notes = [], []
    n = 0
    for obj in objs:
        base = os.path.basename(obj)
        maxilp = base.startswith("variants_") and not base.endswith(".wide.o")
        for name, m in kernels_of(obj):
            n += 1
            is_step = "step_kernel" in name
            production = is_step and re.search(r"ELb[01]ELb0E+vNS_5KArgsE$", name) is not None
            if check:
                if maxilp and (m["vgpr_count"] > 256 or m["agpr_count"] > 0):
                    bad.append(f"{base}: {short(name)} is in the max-ILP group with vgpr_count {m['vgpr_count']}, "
                               f"agpr_count {m['agpr_count']} (limit 256 / 0): move its width to PTRWM_WIDTHS_WIDE")
                if production and m["private_segment_fixed_size"] > 0:
                    msg = f"{base}: production kernel {short(name)} uses {m['private_segment_fixed_size']} B of scratch"
                    (bad if maxilp else notes).append(msg)
            elif flt in name or flt in short(name):
                print(f"{base:34s} {short(name):95s} vgpr {m['vgpr_count']:3d} agpr {m['agpr_count']:2d} vspill "
                      f"{m['vgpr_spill_count']:3d} sgpr {m['sgpr_count']:3d} sspill {m['sgpr_spill_count']:3d} scratch "
                      f"{m['private_segment_fixed_size']:4d} lds {m['group_segment_fixed_size']}")
    if check:
        for msg in notes:
            print("note:", msg)
        if bad:
            print("\n".join(bad))
            sys.exit(f"kernel_stats --check: {len(bad)} violation(s) in {n} kernels")
        print(f"kernel_stats --check: {n} kernels in {len(objs)} objects ok (max-ILP group: <= 256 VGPRs, no AGPRs; "
              "production step kernels: no scratch)")


if __name__ == "__main__":
    main()
