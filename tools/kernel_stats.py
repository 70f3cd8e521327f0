"""Register / spill / LDS figures of the compiled kernels, read from the code-object metadata of the BUILT objects
(rwm-pt-pytorch_amd/build/*.o: .hip_fatbin -> gfx950 code object -> NT_AMDGPU_METADATA), no recompilation.

    python tools/kernel_stats.py [--objdir DIR] [filter-substring]     table of every kernel whose name contains the filter
    python tools/kernel_stats.py --check [--objdir DIR | --objs a.o b.o ...]   the build gate run by csrc/Makefile

--check fails (exit 1) if ANY kernel needs more than 256 VGPRs or any AGPR.  That is the regime (AGPR copies next to
~150-200 SGPRs spilled into VGPR lanes) in which hipcc produced wrong code for the one-thread-per-replica kernels of the
widths 80 / 100 twice: in round 1 under the max-ILP scheduler (profiles/r02_miscompile_width80.txt), in round 2 under the
DEFAULT scheduler after a scheduling fence moved.  Those kernels are retired (dim > 64 runs the lane-split kernel); the
gate keeps any future kernel out of that regime.
It also fails if a PRODUCTION step kernel (FULL = false) of the max-ILP group (variants_*.o) uses ANY scratch memory
(round 2 found and fixed 16 + 4 DP bytes per thread in every HybridRosenbrock kernel this way: an array the optimiser had
made dynamically indexed; until round 4 up to 64 bytes were tolerated - the headline kernel sat at the 128-VGPR cap with
12-32 B of it, part of which was reloaded in every step; now none does: COLD_SCRATCH_BYTES = 0); scratch elsewhere (the 1024-thread lane-split variants of the dim > 64 class spill a dozen VGPRs) is reported;
if a production step kernel spills more SGPRs than PROD_SGPR_SPILL_CEILING (a ratchet: each spilled SGPR is a
v_writelane / v_readlane pair in the step loop, and heavy SGPR spilling next to the empty-asm value barriers is the other
ingredient of the register regime above); and if an object yields NO kernel at all (a different ROCm layout, a stripped
object: the gate must not pass without having looked at anything)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
PROD_SGPR_SPILL_CEILING = 96  # ratchet: 163 when introduced in round 3 (HybridRosenbrock<64> + UniformRadius thread form, now 34); 96 at the end of round 3 = the largest production kernel (a run-time-dim lane-split W = 28 kernel, 94) + the +-2 the count moves by between builds of unrelated changes; lower it when that kernel improves, never raise it
QUAD_SCRATCH_CEILING = 128  # bytes per thread, production lane-split kernels (round 4; a ratchet: lower it, never raise it): worst when introduced 120 B - the double-state (state_f64) twins of the run-time-dim W = 28 kernels, dims 97..104; float kernels: at most 36 B; the 1024-thread class that spilled 204 B inside the step loop is retired
STREAM_SCRATCH_CEILING = 96  # bytes per thread, streaming twins (ratchet): worst when introduced 80 B (RoughCarpet<50> + Normal, at the 256-VGPR gate with one wave per SIMD resident); dims <= 30: none
MAX_KERNELS = 3200  # budget: every (target, proposal, width, twin) is a kernel to build, ship and keep correct (3 099 at the end of round 3, 3 387 with the streaming twins, 3 101 after retiring the 1024-thread lane-split class)
MAX_LIBRARY_BYTES = 88 << 20  # budget for libptrwm_hip.so (--size, run by the Makefile after the link)
STREAM_SGPR_SPILL_CEILING = 128  # the streaming twins (round 4; kernel.h STREAM): their own ratchet - the loop over groups keeps a dozen more scalars alive across the step than the classic kernel's single pass (worst when introduced: Hypercube<50> + UniformRadius, whose verdict is a chain of 64-bit lane masks)
COLD_SCRATCH_BYTES = 0  # production step kernels of the max-ILP group: no scratch at all (64 until round 4, when the headline kernel still parked loop-invariant words - and, it turned out, two values it reloaded every step - there: profiles/r04_scratch_ab.txt)
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
    if "--size" in args:  # python tools/kernel_stats.py --size <libptrwm_hip.so>: the library-size budget (Makefile, after the link)
        path = args[args.index("--size") + 1]
        size = os.path.getsize(path)
        if size > MAX_LIBRARY_BYTES:
            sys.exit(f"kernel_stats --size: {path} is {size / 2**20:.1f} MiB, budget {MAX_LIBRARY_BYTES / 2**20:.0f} MiB")
        print(f"kernel_stats --size: {os.path.basename(path)} {size / 2**20:.1f} MiB (budget {MAX_LIBRARY_BYTES / 2**20:.0f} MiB)")
        return
    check = "--check" in args
    args = [a for a in args if a != "--check"]
    objdir = os.path.join(ROOT, "rwm-pt-pytorch_amd", "build")
    if "--objdir" in args:
        i = args.index("--objdir")
        objdir = args[i + 1]
        del args[i:i + 2]
    objs = None
    if "--objs" in args:  # the Makefile's own object list: stale objects of retired variants in the directory are not looked at
        i = args.index("--objs")
        objs = args[i + 1:]
        del args[i:]
    flt = args[0] if args else ""
    if objs is None:
        objs = sorted(glob.glob(os.path.join(objdir, "*.o")))
    if not objs:
        sys.exit(f"no objects under {objdir}: build first (make -C rwm-pt-pytorch_amd/csrc)")
    bad, notes = [], []
    n = n_step = 0
    worst_spill = (0, "")
    worst_scratch = (0, "")
    for obj in objs:
        base = os.path.basename(obj)
        maxilp = base.startswith("variants_") and not base.endswith(".wide.o")
        found = kernels_of(obj)
        if check and not found:
            bad.append(f"{base}: no gfx950 kernel metadata found (llvm-objcopy / clang-offload-bundler / llvm-readelf under "
                       f"{LLVM} could not read it): the gate has nothing to check")
        for name, m in found:
            n += 1
            is_step = "step_kernel" in name
            # production = FULL false: ptrwm_step_kernel<Target, Proposal, DP, EXACT, FULL, STREAM>,
            # ptrwm_quad_step_kernel<Target, Proposal, W, DEXACT, MAXT, FULL, F64>
            if "quad_step_kernel" in name:
                production = re.search(r"ELb0ELb[01]EEEvNS_5KArgsE$", name) is not None
            else:
                production = is_step and re.search(r"ELb[01]ELb0ELb[01]EEEvNS_5KArgsE$", name) is not None
            n_step += is_step
            if production and m["sgpr_spill_count"] > worst_spill[0] and not (
                    is_step and "quad_step_kernel" not in name and re.search(r"ELb0ELb1EEEvNS_5KArgsE$", name)):
                worst_spill = (m["sgpr_spill_count"], f"{base}: {short(name)}")
            streaming = is_step and re.search(r"ELb0ELb1EEEvNS_5KArgsE$", name) is not None and "quad_step_kernel" not in name
            ceiling = STREAM_SGPR_SPILL_CEILING if streaming else PROD_SGPR_SPILL_CEILING
            if check:
                if production and m["sgpr_spill_count"] > ceiling:
                    bad.append(f"{base}: production kernel {short(name)} spills {m['sgpr_spill_count']} SGPRs "
                               f"(ceiling {ceiling})")
                if m["vgpr_count"] > 256 or m["agpr_count"] > 0:
                    bad.append(f"{base}: {short(name)} needs vgpr_count {m['vgpr_count']}, agpr_count {m['agpr_count']} "
                               "(limit 256 / 0): the register regime hipcc miscompiled twice")
                if production and "quad_step_kernel" in name and m["private_segment_fixed_size"] > QUAD_SCRATCH_CEILING:
                    bad.append(f"{base}: production lane-split kernel {short(name)} uses {m['private_segment_fixed_size']} B of "
                               f"scratch (ceiling {QUAD_SCRATCH_CEILING})")
                if production and m["private_segment_fixed_size"] > 0:
                    msg = f"{base}: production kernel {short(name)} uses {m['private_segment_fixed_size']} B of scratch"
                    # The rule guards the step loop: an array that went to scratch (dynamic indexing: >= 4 x width bytes, found
                    # that way in round 2) or spills inside it.  A few loop-invariant words the allocator parks in scratch at
                    # the register cap - written before the loop, read back in the epilogue or in a swap event - are
                    # tolerated up to COLD_SCRATCH_BYTES; tools/issue_model.py checks that the Metropolis-step path of the
                    # headline kernel contains no scratch store and at most one reload.
                    limit = STREAM_SCRATCH_CEILING if streaming else COLD_SCRATCH_BYTES
                    (bad if (maxilp and m["private_segment_fixed_size"] > limit) else notes).append(msg)
                    if maxilp and not streaming:
                        worst_scratch = max(worst_scratch, (m["private_segment_fixed_size"], f"{base}: {short(name)}"))
            elif flt in name or flt in short(name):
                print(f"{base:34s} {short(name):95s} vgpr {m['vgpr_count']:3d} agpr {m['agpr_count']:2d} vspill "
                      f"{m['vgpr_spill_count']:3d} sgpr {m['sgpr_count']:3d} sspill {m['sgpr_spill_count']:3d} scratch "
                      f"{m['private_segment_fixed_size']:4d} lds {m['group_segment_fixed_size']}")
    if check and n > MAX_KERNELS:
        bad.append(f"{n} kernels in the build, budget {MAX_KERNELS}: every variant axis multiplies build time, library size and "
                   "the surface the oracle checks have to cover - retire something before adding")
    if check:
        if notes:
            print(f"note: {len(notes)} production kernels outside the max-ILP group use scratch, e.g. {notes[0]}")
        if bad:
            print("\n".join(bad))
            sys.exit(f"kernel_stats --check: {len(bad)} violation(s) in {n} kernels")
        print(f"kernel_stats --check: {n} kernels ({n_step} step kernels) in {len(objs)} objects ok (every kernel <= 256 VGPRs "
              f"and no AGPRs; production step kernels: at most {COLD_SCRATCH_BYTES} B of scratch in the max-ILP group "
              f"[worst: {worst_scratch[0]} B, {worst_scratch[1] or 'none'}], at most {worst_spill[0]} spilled "
              f"SGPRs [{worst_spill[1]}], ceiling {PROD_SGPR_SPILL_CEILING})")


if __name__ == "__main__":
    main()
