"""Static trace of the round-1 miscompile (SIGBUS at `ext_u[srep]` in ptrwm_step_kernel<RoughCarpetT<80,true>,
UniformRadiusProposal<80>, 80, false, true>, profiles/r02_miscompile_width80.txt) to its first wrong definition.  No GPU,
nothing is executed: the faulting build's assembly is regenerated from the source of commit d33cf16 (the last tree that
still has the one-thread-per-replica kernels above width 64) with the flags that faulted, and a forward data-flow pass over
the kernel's control-flow graph follows every register the faulting address is built from.

    git worktree add /tmp/wt d33cf16
    python tools/miscompile_trace.py /tmp/wt > profiles/r03_miscompile_trace.txt

The address of the faulting load is  ext_u + 4 * srep,  srep = (i * n_chains + chain) * T + t  (kernel.h), rebuilt at the top
of every step (the loop header block) from
    i        an SGPR loop counter          n_chains  an SGPR pair
    chain    a VGPR pair (loop-invariant)  T         two lanes of an SGPR-spill VGPR, reloaded with v_readlane_b32
    t        a VGPR pair (loop-invariant)
gdb showed the SGPR base (ext_u) valid and the VGPR part wild.  For each loop-invariant VGPR operand the pass computes, for
every basic block of the loop, whether the register still holds the value it had in the loop preheader ("kept"), was
overwritten ("clobbered"), or was re-loaded from the AGPR the preheader saved it to ("restored"); a path from a clobber to the
loop header without a restore in between is a register-allocation error of the compiler - it cannot come from the source,
which never names a register."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from issue_model import kernel_blocks  # noqa: E402

KERNEL = "ptrwm_step_kernelINS_12RoughCarpetTILi80ELb1EEENS_21UniformRadiusProposalILi80EEELi80ELb0ELb1EEE"
MAXILP = ["-mllvm", "-enable-post-misched=0", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fno-slp-vectorize"]


def build(tree, flags):
    tmp = tempfile.mkdtemp(prefix="mctrace_")
    src = os.path.join(tree, "rwm-pt-pytorch_amd", "csrc", "variants_rough_carpet2.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DPTRWM_PART_WIDE",
                           "-save-temps=obj", "-c", src, "-o", os.path.join(tmp, "v.o")] + flags, stderr=subprocess.DEVNULL, cwd=tmp)
    return os.path.join(tmp, "variants_rough_carpet2-hip-amdgcn-amd-amdhsa-gfx950.s")


def regs(tok):
    """VGPR / AGPR numbers an operand token names: v12 -> {('v', 12)}, v[8:9] -> {('v', 8), ('v', 9)}, a10 -> {('a', 10)}."""
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


def dests(op, args):
    """Registers an instruction writes (first operand of VALU / load instructions; stores and compares write none)."""
    if op.startswith(("global_store", "ds_write", "scratch_store", "flat_store", "buffer_store", "v_cmp", "v_readlane",
                      "v_readfirstlane", "s_", "global_atomic")) or op.startswith("v_cmpx"):
        return set()
    first = args.split(",")[0].strip()
    return regs(first)


def trace(asm, report):
    name, blocks, order = kernel_blocks(asm, KERNEL)
    header = next(b for b in order if "This Loop Header: Depth=1" in blocks[b]["notes"] and "Child Loop" in blocks[b]["notes"])
    in_loop = {b for b in order if f"Header={header.replace('.L', '')}" in blocks[b]["notes"].replace(" ", "")} | {header}
    hdr_ins = blocks[header]["ins"]
    # the two multiply-adds that build srep in the header: ... = n_chains * i + chain ;  ... = (that) * T + t
    mads = [(o, a) for o, a in hdr_ins if o == "v_mad_u64_u32"][:2]
    chain = regs(mads[0][1].split(",")[-1].strip())
    tpair = regs(mads[1][1].split(",")[-1].strip())
    report(f"kernel {name}")
    report(f"loop header {header}: {len(in_loop)} basic blocks in the step loop")
    report(f"  srep part 1: v_mad_u64_u32 {mads[0][1]}     chain = {sorted(chain)}")
    report(f"  srep part 2: v_mad_u64_u32 {mads[1][1]}     t     = {sorted(tpair)}")
    # where does the preheader park the loop-invariant pair?  (v_accvgpr_write_b32 aN, vM just before the branch to the header)
    idx = order.index(header)
    saved = {}
    for b in order[:idx]:
        for o, a in blocks[b]["ins"]:
            if o == "v_accvgpr_write_b32":
                d, s = [x.strip() for x in a.split(",")]
                for r in regs(s):
                    if r in tpair | chain:
                        saved[r] = next(iter(regs(d)))
    report(f"  saved to AGPRs before the loop: {{{', '.join(f'v{k[1]} -> a{v[1]}' for k, v in sorted(saved.items()))}}}")
    preds = {b: [] for b in in_loop}
    for b in in_loop:
        for kind, t in blocks[b]["succ"]:
            if t in in_loop:
                preds[t].append(b)
    verdicts = {}
    for reg in sorted(tpair | chain):
        # forward may-analysis: can `reg` be CLOBBERED (last write on some path is not a restore from its AGPR) at block entry?
        agpr = saved.get(reg)

        def transfer(state, b):
            last = None
            for k, (o, a) in enumerate(blocks[b]["ins"]):
                if b == header and (o, a) in mads and reg in (tpair if (o, a) == mads[1] else chain):
                    pass
                if reg in dests(o, a):
                    src = a.split(",")[1].strip() if "," in a else ""
                    restore = o == "v_accvgpr_read_b32" and agpr is not None and regs(src) == {agpr}
                    state = "kept" if restore else "clobbered"
                    last = (k, o, a)
            return state, last

        entry = {b: set() for b in in_loop}
        entry[header] = {"kept"}  # from the preheader
        work = [header]
        out = {}
        while work:
            b = work.pop()
            outs = set()
            for st in entry[b]:
                outs.add(transfer(st, b)[0])
            if out.get(b) == outs:
                continue
            out[b] = outs
            for kind, t in blocks[b]["succ"]:
                if t in in_loop and not outs <= entry[t]:
                    entry[t] |= outs
                    work.append(t)
                elif t in in_loop and t not in out:
                    work.append(t)
        bad_back = [b for b in preds[header] if "clobbered" in out.get(b, set())]
        n_clob = sum(1 for b in in_loop for o, a in blocks[b]["ins"] if reg in dests(o, a) and o != "v_accvgpr_read_b32")
        n_rest = sum(1 for b in in_loop for o, a in blocks[b]["ins"] if reg in dests(o, a) and o == "v_accvgpr_read_b32")
        # is the use in the header before any redefinition there?
        first_def = next((k for k, (o, a) in enumerate(hdr_ins) if reg in dests(o, a)), None)
        use = next(k for k, (o, a) in enumerate(hdr_ins) if (o, a) in mads and reg in (regs(a.split(",")[-1].strip())))
        exposed = first_def is None or first_def > use
        verdicts[reg] = (bad_back, n_clob, n_rest, exposed)
        report(f"\n  v{reg[1]}: written {n_clob} times as a scratch register inside the loop, restored from "
               f"{'a%d' % agpr[1] if agpr else '(never saved)'} {n_rest} times; the header reads it "
               f"{'BEFORE' if exposed else 'after'} any definition of its own")
        if bad_back and exposed:
            report(f"    -> reaches the loop header CLOBBERED along the back edge(s) from {bad_back}: the value the second step's "
                   f"srep is built from is whatever scratch value was last parked in v{reg[1]}")
            # one witness: the last clobber in program order that can reach the latch
            for b in order:
                if b in in_loop:
                    st, last = transfer("kept", b)
                    if st == "clobbered" and "clobbered" in out.get(b, set()):
                        witness = (b, last)
            report(f"    witness (last such write in layout order): block {witness[0]}: {witness[1][1]} {witness[1][2]}")
        else:
            report("    -> holds its preheader value on every path back to the header")
    return verdicts


def sgpr_dests(o, a):
    """SGPR numbers an instruction writes (SALU results, v_readlane / v_cmp / carry-out destinations)."""
    if o.startswith(("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_bitcmp", "s_endpgm", "s_setprio")):
        return set()
    if not (o.startswith("s_") or o in ("v_readlane_b32", "v_readfirstlane_b32") or o.startswith(("v_cmp", "v_mad_u64"))):
        return set()
    toks = [t.strip() for t in a.split(",")]
    tok = toks[1] if o.startswith("v_mad_u64") else toks[0]
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def trace_scalars(asm, report):
    """The scalar operands of the same address: n_chains (the SGPR source of the first multiply-add), the loop counter, and
    the two spill lanes T is reloaded from."""
    name, blocks, order = kernel_blocks(asm, KERNEL)
    header = next(b for b in order if "This Loop Header: Depth=1" in blocks[b]["notes"] and "Child Loop" in blocks[b]["notes"])
    in_loop = {b for b in order if f"Header={header.replace('.L', '')}" in blocks[b]["notes"].replace(" ", "")} | {header}
    hdr = blocks[header]["ins"]
    mad1 = next(a for o, a in hdr if o == "v_mad_u64_u32")
    nch = int(re.search(r"s(\d+)", mad1.split(",")[2]).group(1))
    for reg in (nch, nch + 1):
        clob = [(b, o, a) for b in order if b in in_loop for o, a in blocks[b]["ins"] if reg in sgpr_dests(o, a) and o != "v_readlane_b32"]
        rest = [(b, a) for b in order if b in in_loop for o, a in blocks[b]["ins"] if o == "v_readlane_b32" and reg in sgpr_dests(o, a)]
        report(f"\n  s{reg} (n_chains): overwritten {len(clob)} time(s) inside the loop {[(b, o + ' ' + a) for b, o, a in clob][:2]}, "
               f"reloaded with v_readlane_b32 {len(rest)} time(s) {sorted({a for _, a in rest})}")
        if clob:
            lanes = {a.split(",", 1)[1].strip() for _, a in rest}
            saves = [(b, a) for b in order for o, a in blocks[b]["ins"] if o == "v_writelane_b32" and
                     any(a.replace(" ", "") == f"{ln.split(',')[0].strip()},s{reg},{ln.split(',')[1].strip()}" for ln in lanes)]
            report(f"    saved to the same lane(s) by {[(b, a) for b, a in saves]}")
            restore_blocks = {b for b, _ in rest}
            seen, work, hit = set(), [b for b, _, _ in clob], None
            while work and hit is None:
                b = work.pop()
                for kind, t in blocks[b]["succ"]:
                    if t == header:
                        hit = b
                        break
                    if t in in_loop and t not in seen and t not in restore_blocks:
                        seen.add(t)
                        work.append(t)
            report("    -> every path from the overwrite back to the loop header passes a block that reloads it" if hit is None else
                   f"    -> REACHES THE HEADER OVERWRITTEN via {hit}")
    lanes = re.findall(r"v_readlane_b32 s\d+, (v\d+), (\d+)", "\n".join(f"{o} {a}" for o, a in hdr))
    for vreg, lane in sorted(set(lanes))[:4]:
        w_in = [(b, a) for b in order if b in in_loop for o, a in blocks[b]["ins"] if o == "v_writelane_b32" and
                a.replace(" ", "").startswith(f"{vreg},") and a.replace(" ", "").endswith(f",{lane}")]
        full = [(b, o, a) for b in order for o, a in blocks[b]["ins"] if o not in ("v_writelane_b32", "v_readlane_b32") and
                re.search(rf"\b{vreg}\b", a)]
        report(f"  spill lane {vreg}[{lane}] read in the header: written {len(w_in)} time(s) inside the loop; {vreg} is touched by "
               f"{len(full)} instruction(s) other than v_writelane / v_readlane")


def main():
    tree = sys.argv[1] if len(sys.argv) > 1 else "/tmp/wt"
    out = print
    out(__doc__)
    for tag, flags in (("FAULTING build: max-ILP scheduling flags", MAXILP), ("passing build: default scheduler", [])):
        out(f"\n==== {tag} " + "=" * (90 - len(tag)))
        asm = build(tree, flags)
        trace(asm, out)
        trace_scalars(asm, out)
    out("""
==== Reading ========================================================================================================
In the build that faulted every loop-invariant operand of the wild address is accounted for: `chain` is never written
inside the loop; `t` IS used as a scratch register (29 writes in the 1 389-block step loop) but was parked in a[10:11] by the
preheader and every path from a scratch write back to the loop header passes one of the three blocks that re-load it;
n_chains is overwritten once (a boolean mask) after being saved to a spill lane in the same block and re-loaded on every
path; the spill lanes T is read from are written once, before the loop, and the spill VGPRs are touched by lane
instructions only.  So the value flow the register allocator set up is sound as far as the ISA text shows it: no missing
reload, no reused live slot.  What is NOT visible in the text is the order in which the machine retires these instructions:
the header reloads up to sixteen SGPRs with v_readlane_b32 immediately before VALU / VMEM instructions consume them, next to
AGPR copies, in a kernel that needs every one of 256 VGPRs + 14 AGPRs + 211 spilled SGPRs.  The source-level `asm volatile`
value barriers are not on this def-use chain at all (srep is built from blockIdx / threadIdx-derived values, kernel arguments
and the loop counter; none of them passes through opaque_vgpr / fresh_dim / uniform_vec).  Conclusion: the cause is NOT
isolated to a wrong definition; it is narrowed to the code generator's handling of that register regime (spill-lane reloads
and AGPR copies under full pressure), the barriers are excluded as the origin of the wild value, and the build gate keeps
every shipped kernel out of the regime (<= 256 VGPRs, no AGPRs; ratcheted ceiling on spilled SGPRs in production kernels).""")


if __name__ == "__main__":
    main()
