"""Cost-weighted issue roofline of a fused step kernel: the opcode histogram of ONE Metropolis step of the kernel (the path
through its step loop that a step without a swap event takes), each opcode weighted with its measured issue cost
(tools/issue_cost.hip -> profiles/r03_issue_costs.json), against the SIMD time a wave-step really takes.

    python tools/issue_model.py [--tu rough_carpet2] [--kernel <mangled-name filter>] [--costs profiles/r03_issue_costs.json]
                                [--valu-per-wave-step N] [--salu-per-wave-step N] [--ns-per-wave-step X] [--json out.json]

How the histogram is obtained (no GPU needed, ~1 min): the translation unit is compiled again with the Makefile's flags and
-save-temps; the annotated assembly gives the basic blocks of the kernel and its step loop (the depth-1 loop holding the
Philox multiplies).  A step WITHOUT a swap event executes one path from the loop header back to itself; that path is found by
enumerating the simple paths of the loop's control-flow graph and taking the one that avoids LDS traffic, barriers, inner
loops (all three are the swap event) and in-loop scalar loads of a NULL parameter vector (the `scaled` twin of the target):
`s_cbranch_exec*` skips are not taken (the wave has live lanes).  The choice is CHECKED, not trusted: the path's VALU and SALU
instruction counts must equal the SQ_INSTS_VALU / SQ_INSTS_SALU per wave-step that rocprofv3 counted for the same kernel run
without swap events (bench.py --swap-every 1073741824, profiles/r03_pmc_cfg3_noswap.csv) when those are given.

Output: per opcode count x cost, the sum = busy SIMD time per wave-step, and its ratio to the measured time per wave-step
(kernel time x SIMDs / (waves x steps)) = the cost-weighted issue fraction.  Opcodes without a measured cost are priced
as v_fma_f32 and listed.

Round 4 (profiles/r04_issue_costs.json: costs in CYCLES of the clock measured inside each microbenchmark launch, at eight
waves per SIMD = saturated issue, tools/residency_probe.hip having shown what "N waves per SIMD" really is): every VALU
instruction on the path is priced by its opcode AND by whether it reads an SGPR - on gfx950 a full-rate opcode (2.3-2.4
cycles per wave64 instruction: add / mul / fma / xor / shifts / bitop3 on VGPRs, inline or literal constants) issues at HALF
rate (4.1-4.2) as soon as one source is an SGPR, the same rate as the inherently half-rate opcodes (max3 / med3 / and_or /
cndmask / cvt / mad_u64 / mul_lo / compares / lane ops / fp64); transcendentals 8.1.  The sum is the issue-time FLOOR of the
kernel's own instruction mix; measured cycles per wave-step (kernel time x clock x SIMDs / (waves x steps)) over that
floor says how far the kernel is from it.  `--costs` in the old (ns) format still works."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCHED = "-mllvm -enable-post-misched=0 -mllvm -amdgpu-sched-strategy=max-ilp -fno-slp-vectorize".split()
HEADLINE = "ptrwm_step_kernelINS_12RoughCarpetTILi30ELb1EEENS_14NormalProposalILi30EEELi30ELb1ELb0ELb0EEE"

# opcode (as printed, VOP suffixes stripped) -> the measured opcode whose cost it shares
ALIAS = {
    "v_sub_f32": "v_add_f32", "v_subrev_f32": "v_add_f32", "v_max_f32": "v_add_f32", "v_min_f32": "v_add_f32",
    "v_mov_b32": "v_add_f32", "v_and_b32": "v_xor_b32", "v_or_b32": "v_xor_b32", "v_lshlrev_b32": "v_lshrrev_b32",
    "v_fmac_f32": "v_fma_f32", "v_fmaak_f32": "v_fmamk_f32", "v_min3_f32": "v_max3_f32", "v_sub_u32": "v_add_u32",
    "v_subrev_u32": "v_add_u32", "v_add3_u32": "v_and_or_b32", "v_cmp_gt_f32": "v_cmp_lt_f32", "v_cmp_ge_f32": "v_cmp_lt_f32",
    "v_cmp_le_f32": "v_cmp_lt_f32", "v_cmp_ngt_f32": "v_cmp_lt_f32", "v_cmp_nlt_f32": "v_cmp_lt_f32", "v_cmp_eq_u32": "v_cmp_lt_f32",
    "v_cmp_ne_u32": "v_cmp_lt_f32", "v_cmp_lt_u32": "v_cmp_lt_f32", "v_cmp_gt_u32": "v_cmp_lt_f32", "v_cmp_lt_i32": "v_cmp_lt_f32",
    "v_cmp_gt_i32": "v_cmp_lt_f32", "v_cmp_class_f32": "v_cmp_lt_f32", "v_cvt_f32_i32": "v_cvt_f32_u32", "v_mul_hi_u32": "v_mul_lo_u32",
    "v_mul_f64": "v_fma_f64", "v_accvgpr_read_b32": "v_add_f32", "v_accvgpr_write_b32": "v_add_f32", "v_rsq_f32": "v_rcp_f32",
    "v_ldexp_f32": "v_mul_f32", "v_bfe_u32": "v_and_or_b32", "v_lshl_add_u32": "v_and_or_b32", "v_lshl_or_b32": "v_and_or_b32",
    "v_xad_u32": "v_and_or_b32", "v_add_co_u32": "v_add_u32", "v_addc_co_u32": "v_add_u32", "v_mul_u32_u24": "v_add_u32",
    "v_mad_u32_u24": "v_and_or_b32", "v_readfirstlane_b32": "v_readlane_b32", "v_mul_legacy_f32": "v_mul_f32",
}


def strip(op):
    return re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)


def compile_tu(tu, flags):
    tmp = tempfile.mkdtemp(prefix="issue_model_")
    prefix = "quad_" if tu.startswith("quad_") else "variants_"
    src = os.path.join(ROOT, "rwm-pt-pytorch_amd", "csrc", f"{prefix}{tu.replace('quad_', '')}.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                           "-save-temps=obj", "-c", src, "-o", os.path.join(tmp, "v.o")] + flags, stderr=subprocess.DEVNULL, cwd=tmp)
    return os.path.join(tmp, os.path.basename(src).replace(".hip", "-hip-amdgcn-amd-amdhsa-gfx950.s"))


def kernel_blocks(asm_path, flt):
    lines = open(asm_path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN5ptrwm") and flt in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    name = lines[start].split(":")[0]
    blocks, order, cur = {}, [], None
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %bb\.(\d+):", l)
        if m:
            cur = m.group(1) if l.startswith(".L") else f"bb.{m.group(1)}"
            blocks[cur] = {"ins": [], "notes": l, "succ": []}
            order.append(cur)
            continue
        if cur is None:
            continue
        if l.lstrip().startswith(";"):
            blocks[cur]["notes"] += " " + l.strip()
            continue
        m = re.match(r"\s+([a-z][a-z0-9_]+)\s*(.*)", l)
        if m and not m.group(1).startswith("."):
            blocks[cur]["ins"].append((m.group(1), m.group(2)))
    for i, b in enumerate(order):
        ins = blocks[b]["ins"]
        nxt = order[i + 1] if i + 1 < len(order) else None
        fall = True
        for op, args in ins:
            if op.startswith("s_cbranch"):
                blocks[b]["succ"].append((op, args.split()[0]))
            elif op == "s_branch":
                blocks[b]["succ"].append((op, args.split()[0]))
                fall = False
            elif op in ("s_endpgm", "s_setpc_b64"):
                fall = False
        if fall and nxt:
            blocks[b]["succ"].append(("fall", nxt))
    return name, blocks, order


def features(b):
    ops = [o for o, _ in b["ins"]]
    return {
        # (LDS traffic marks the swap event - except the no-return ds_add_f64 that every counted step adds its squared jump with)
        "lds": sum(o.startswith("ds_") and not o.startswith("ds_add_f64") for o in ops), "barrier": sum(o == "s_barrier" for o in ops),
        "sload": sum(o.startswith("s_load") for o in ops), "inner": int("Depth=2" in b["notes"] or "Depth 2" in b["notes"] and "Child" not in b["notes"]),
        "global": sum(o.startswith(("global_", "flat_", "buffer_", "scratch_")) for o in ops),
        "cold": int("ptrwm-cold-path" in b["notes"]),  # PTRWM_COLD_PATH() of philox.h
    }


def step_path(blocks, order):
    loops = [b for b in order if "This Loop Header: Depth=1" in blocks[b]["notes"]]
    header = max(loops, key=lambda b: sum(o == "v_mad_u64_u32" for o, _ in blocks[b]["ins"]) * 1000 + len(blocks[b]["ins"]))
    in_loop = {b for b in order if f"Header={header.replace('.L', '')}" in blocks[b]["notes"].replace(" ", "")} | {header}
    # Among the paths from the header back to itself that avoid the swap event: least penalty, then most VALU work.  The
    # loop body without its back edge and without inner loops is a DAG: one memoised pass (generic kernels have > 1000 blocks).
    valu = {b: sum(o.startswith("v_") for o, _ in blocks[b]["ins"]) for b in in_loop}
    INF = (10**9, 0)
    memo, visiting = {}, set()

    def cost(t):
        f = features(blocks[t])
        return 10 * f["lds"] + 1000 * f["barrier"] + 3 * f["sload"] + 1000 * f["inner"] + 5 * f["global"] + 100 * f["cold"]

    def best_from(b):
        """(penalty, -valu, next block) of the best continuation from the END of block b to the header."""
        if b in memo:
            return memo[b]
        if b in visiting:
            return (INF[0], 0, None)
        visiting.add(b)
        out = (INF[0], 0, None)
        falls_cold = any(k == "fall" and features(blocks[t])["cold"] for k, t in blocks[b]["succ"])
        for kind, t in blocks[b]["succ"]:
            if kind.startswith("s_cbranch_exec") and not falls_cold:
                continue  # a skip over a block for waves without live lanes: not taken (but taken over a cold block)
            if t == header:
                cand = (0, 0, header)
            elif t in in_loop:
                sub = best_from(t)
                if sub[2] is None:
                    continue
                cand = (cost(t) + sub[0], -valu[t] + sub[1], t)
            else:
                continue
            if (cand[0], cand[1]) < (out[0], out[1]):
                out = cand
        visiting.discard(b)
        memo[b] = out
        return out

    import sys as _sys
    _sys.setrecursionlimit(20000)
    best = [None, None]
    first = best_from(header)
    if first[2] is not None:
        path, b = [header], header
        while True:
            nxt = best_from(b)[2]
            if nxt is None or nxt == header:
                break
            path.append(nxt)
            b = nxt
        best = [(first[0], first[1]), path]
    if best[1] is None:
        sys.exit("no path from the step-loop header back to itself found")
    return header, best[1], best[0][0]


TRANS = ("v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f32", "v_rsq_f32")
# opcodes that issue at half rate whatever their operands (measured: tools/issue_cost.hip) and the measured key that prices them
HALF = {"v_max3_f32": "v_max3_f32", "v_min3_f32": "v_max3_f32", "v_med3_f32": "v_med3_f32", "v_and_or_b32": "v_and_or_b32",
        "v_cndmask_b32": "v_cndmask_b32", "v_cvt_f32_u32": "v_cvt_f32_u32", "v_cvt_f32_i32": "v_cvt_f32_u32",
        "v_cvt_u32_f32": "v_cvt_f32_u32", "v_mad_u64_u32": "v_mad_u64_u32", "v_mul_lo_u32": "v_mul_lo_u32",
        "v_mul_hi_u32": "v_mul_hi_u32 v,v,v", "v_cvt_f64_f32": "v_cvt_f64_f32", "v_add_f64": "v_add_f64", "v_mul_f64": "v_fma_f64",
        "v_fma_f64": "v_fma_f64", "v_readlane_b32": "v_readlane_b32", "v_readfirstlane_b32": "v_readlane_b32",
        "v_writelane_b32": "v_writelane_b32", "v_mov_b32_dpp": "v_mov_b32_dpp", "v_lshl_add_u32": "v_and_or_b32",
        "v_add3_u32": "v_and_or_b32", "v_lshl_or_b32": "v_and_or_b32", "v_bfe_u32": "v_and_or_b32", "v_xad_u32": "v_and_or_b32",
        "v_mad_u32_u24": "v_and_or_b32", "v_lshl_add_u64": "v_add_f64", "v_mbcnt_lo_u32_b32": "v_and_or_b32",
        "v_mbcnt_hi_u32_b32": "v_and_or_b32", "v_pk_fma_f32": "v_pk_fma_f32", "v_pk_mul_f32": "v_pk_mul_f32",
        "v_pk_add_f32": "v_pk_add_f32"}
# full-rate opcodes: the measured key of the all-VGPR form; with an SGPR source they cost SGPR_KEY
FULL = {"v_add_f32": "v_add_f32", "v_sub_f32": "v_add_f32", "v_subrev_f32": "v_add_f32", "v_max_f32": "v_add_f32",
        "v_min_f32": "v_add_f32", "v_mul_f32": "v_mul_f32", "v_mul_legacy_f32": "v_mul_f32", "v_ldexp_f32": "v_mul_f32",
        "v_fma_f32": "v_fma_f32", "v_fmac_f32": "v_fmac_f32 v,v1,v2", "v_fmamk_f32": "v_fmamk_f32", "v_fmaak_f32": "v_fmamk_f32",
        "v_xor_b32": "v_xor_b32", "v_and_b32": "v_and_b32 v,v,v", "v_or_b32": "v_xor_b32", "v_not_b32": "v_xor_b32",
        "v_lshrrev_b32": "v_lshrrev_b32", "v_lshlrev_b32": "v_lshrrev_b32", "v_ashrrev_i32": "v_lshrrev_b32",
        "v_add_u32": "v_add_u32", "v_sub_u32": "v_add_u32", "v_subrev_u32": "v_add_u32", "v_add_co_u32": "v_add_u32",
        "v_addc_co_u32": "v_add_u32", "v_mul_u32_u24": "v_add_u32", "v_bitop3_b32": "v_bitop3_b32 v,v,v1,v2 (accumulate)",
        "v_mov_b32": "v_add_f32", "v_accvgpr_read_b32": "v_add_f32", "v_accvgpr_write_b32": "v_add_f32"}
SGPR_KEYS = ("v_add_f32 v,s,v", "v_fma_f32 v,v,s,s", "v_xor_b32 v,s,v", "v_mul_f32 v,s,v", "v_mov_b32 v,s", "v_bitop3_b32")


def reads_sgpr(op, args):
    """Does this VALU instruction read an SGPR (or VCC / EXEC) as a SOURCE operand?"""
    parts = [p.strip() for p in re.split(r",(?![^\[]*\])", args.split(" bitop3:")[0].split(" quad_perm")[0])]
    if op.startswith("v_cmp"):
        srcs = parts[1:] if re.match(r"^(s\[|vcc|s\d)", parts[0]) else parts
    elif op.startswith("v_mad_u64_u32") or op.startswith("v_add_co") or op.startswith("v_addc_co"):
        srcs = parts[2:]
    else:
        srcs = parts[1:]
    return any(re.match(r"^[-|]*(s\d+|s\[|vcc|exec|ttmp|m0)", x) for x in srcs)


def main_cycles(a, name, header, path, pen, blocks, ops, valu, salu, ok, costs, lib_sha):
    """Round-4 pricing: cycles at saturated issue, by opcode class and operand kind (module docstring)."""
    cyc = {k: v["cycles"] for k, v in costs.items()}
    sgpr_cost = sum(cyc[k] for k in SGPR_KEYS) / len(SGPR_KEYS)
    hist = collections.Counter()  # (opcode, class) -> count
    sequence = []  # the step path in program order: [opcode, class] (tools/mix_probe.py replays it without dependencies)
    for b in path:
        for o, args in blocks[b]["ins"]:
            o = strip(o)
            if not o.startswith("v_"):
                if o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop", "s_load", "s_buffer", "s_cbranch", "s_branch", "s_barrier")):
                    sequence.append([o, "salu"])
                continue
            if o in TRANS:
                cls = "quarter"
            elif o in HALF or o.startswith("v_cmp"):
                cls = "half"
            elif o in FULL:
                cls = "full+sgpr" if reads_sgpr(o, args) else "full"
            else:
                cls = "unpriced"
            hist[(o, cls)] += 1
            sequence.append([o, cls, args])
    rows, floor, by_class, unpriced = [], 0.0, collections.Counter(), []
    for (o, cls), c in hist.most_common():
        if cls == "quarter":
            each = cyc.get(o, cyc["v_exp_f32"])
        elif cls == "half":
            each = cyc[HALF[o]] if o in HALF else cyc["v_cmp_lt_f32"]
        elif cls == "full":
            each = cyc[FULL[o]]
        elif cls == "full+sgpr":
            each = sgpr_cost
        else:
            each = cyc["v_max3_f32"]  # an opcode nobody measured: priced as a half-rate one, and listed
            unpriced.append(o)
        rows.append((o, cls, c, each))
        floor += c * each
        by_class[cls] += c
    print(f"\n{'opcode':24s} {'class':10s} {'count':>6s} {'cycles':>7s} {'sum':>8s} {'share':>6s}")
    for o, cls, c, each in rows:
        print(f"{o:24s} {cls:10s} {c:6d} {each:7.2f} {c * each:8.1f} {c * each / floor:6.1%}")
    n = sum(by_class.values())
    print(f"VALU instructions on the step path: {n}: " + ", ".join(f"{k} {v} ({v / n:.0%})" for k, v in by_class.most_common())
          + (f"; unpriced (as half rate): {sorted(set(unpriced))}" if unpriced else ""))
    print(f"issue-time floor of this instruction mix: {floor:.0f} cycles per wave-step = {floor / n:.2f} cycles per VALU instruction "
          f"(nominal peak: 2.00; saturated full-rate opcode measured: {cyc['v_fma_f32']:.2f})")
    out = {"kernel": name, "path": path, "valu_on_path": valu, "salu_on_path": salu,
           "histogram_by_class": {f"{o} [{cls}]": c for (o, cls), c in hist.items()}, "class_counts": dict(by_class),
           "floor_cycles_per_wave_step": floor, "floor_cycles_per_valu_instruction": floor / n, "unpriced": sorted(set(unpriced)),
           "pmc_check": {"valu_per_wave_step": a.valu_per_wave_step, "salu_per_wave_step": a.salu_per_wave_step, "ok": ok},
           "costs": os.path.relpath(a.costs, ROOT), "waves": a.waves, "lib_sha256": lib_sha, "sequence": sequence}
    if a.ns_per_wave_step and a.clock_ghz:
        meas = a.ns_per_wave_step * a.clock_ghz
        print(f"measured: {a.ns_per_wave_step:.1f} ns of SIMD time per wave-step x {a.clock_ghz:.3f} GHz = {meas:.0f} cycles = "
              f"{meas / n:.2f} per VALU instruction  ->  floor / measured = {floor / meas:.3f}; nominal-peak fraction = {2.0 * n / meas:.3f}")
        out.update({"ns_per_wave_step": a.ns_per_wave_step, "clock_ghz": a.clock_ghz, "measured_cycles_per_wave_step": meas,
                    "frac_of_measured_mix": floor / meas, "frac_of_nominal_peak": 2.0 * n / meas})
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)
    sys.exit(0 if ok else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tu", default="rough_carpet2")
    ap.add_argument("--kernel", default=HEADLINE)
    ap.add_argument("--costs", default=os.path.join(ROOT, "profiles", "r04_issue_costs.json"))
    ap.add_argument("--waves", default="waves_per_simd_8")
    ap.add_argument("--clock-ghz", type=float, help="shader clock of the measured kernel run (GRBM_GUI_ACTIVE / 8 / kernel time)")
    ap.add_argument("--valu-per-wave-step", type=float)
    ap.add_argument("--salu-per-wave-step", type=float)
    ap.add_argument("--ns-per-wave-step", type=float, help="measured: kernel time x SIMDs / (waves x steps)")
    ap.add_argument("--flags", default=None, help="compile flags of the TU's group (default: the Makefile's max-ILP group)")
    ap.add_argument("--asm", default=None, help="an annotated .s to reuse instead of compiling")
    ap.add_argument("--json", default=None)
    ap.add_argument("--from-traffic", default=None, help="key of profiles/traffic.json to take the PMC counts and the kernel time "
                                                         "of (a launch without swap events), e.g. pt_d30_T32_C65536_noswap")
    a = ap.parse_args()
    lib_sha = None
    if a.from_traffic:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[a.from_traffic]
        waves, steps = rec["waves_per_launch"], rec["mh_steps_per_launch"]
        a.valu_per_wave_step = rec["valu_insts_per_launch"] / (waves * steps)
        a.salu_per_wave_step = rec["salu_insts_per_launch"] / (waves * steps)
        a.ns_per_wave_step = rec["profiled_kernel_ms"] * 1e6 * 1024 / (waves * steps)  # 1024 SIMDs
        lib_sha = rec["lib_sha256"]
        a.clock_ghz = a.clock_ghz or rec.get("shader_clock_ghz")
        print(f"PMC record {a.from_traffic}: {waves:.0f} waves x {steps} steps per launch, {rec['profiled_kernel_ms']:.3f} ms")
    asm = a.asm or compile_tu(a.tu, a.flags.split() if a.flags is not None else SCHED)
    name, blocks, order = kernel_blocks(asm, a.kernel)
    header, path, pen = step_path(blocks, order)
    ops = collections.Counter()
    for b in path:
        for o, _ in blocks[b]["ins"]:
            ops[strip(o)] += 1
    valu = sum(c for o, c in ops.items() if o.startswith("v_"))
    # (SQ_INSTS_SALU does not count branches, waits, nops or scalar loads)
    salu = sum(c for o, c in ops.items() if o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop", "s_load", "s_buffer", "s_cbranch", "s_branch", "s_barrier")))
    smem = sum(c for o, c in ops.items() if o.startswith(("s_load", "s_buffer")))
    print(f"kernel {name}\nstep loop header {header}; path of a step without a swap event: {' '.join(path)} (penalty {pen})")
    print(f"instructions on the path: {sum(ops.values())} = {valu} VALU + {salu} SALU + {smem} scalar loads + "
          f"{sum(c for o, c in ops.items() if o.startswith('ds_'))} LDS + {ops.get('s_nop', 0)} s_nop + {ops.get('s_waitcnt', 0)} s_waitcnt + "
          f"{sum(c for o, c in ops.items() if o.startswith('scratch_'))} scratch")
    # spill traffic on the step path: at most one reload of a loop-invariant value (the headline kernel at its 128-VGPR cap
    # reloads one 8-byte Philox product per step from cache - A/B-timed against the arrangement without it,
    # profiles/r03_scratch_ab.txt); a store, or more than one reload, is a regression
    scratch_ops = {o: c for o, c in ops.items() if o.startswith("scratch_")}
    ok = sum(scratch_ops.values()) <= 1 and not any(o.startswith("scratch_store") for o in scratch_ops)
    for what, got, want in (("VALU", valu, a.valu_per_wave_step), ("SALU", salu, a.salu_per_wave_step)):
        if want:
            rel = got / want - 1
            print(f"check against the PMC count without swap events: {what} {got} static vs {want:.1f} counted per wave-step ({rel:+.2%})")
            ok &= abs(rel) < (0.01 if what == "VALU" else 0.05)
    costs = json.load(open(a.costs))[a.waves] if os.path.exists(a.costs) else {}
    if costs and isinstance(next(iter(costs.values())), dict):
        return main_cycles(a, name, header, path, pen, blocks, ops, valu, salu, ok, costs, lib_sha)
    base = costs.get("v_fma_f32")
    rows, busy, unpriced = [], 0.0, []
    for o, c in ops.most_common():
        if not o.startswith("v_"):
            continue
        key = o if o in costs else ALIAS.get(o)
        ns = costs.get(key) if key else None
        if ns is None:
            unpriced.append(o)
            ns = base
        if ns is not None:
            busy += c * ns
        rows.append((o, c, ns))
    # Philox in context.  v_mad_u64_u32 and v_bitop3_b32 timed alone run slower than the same pair issued among full-rate
    # float work of other waves (tools/issue_cost.hip "mix:" probes) - and in the kernel they always are: four waves share
    # a SIMD and three quarters of their instructions are plain fp32.  The in-context price of a (mad, bitop3) pair is the
    # mix of one of each with two fmas, minus the two fmas.
    busy_ctx = None
    mixk = "mix:v_mad_u64_u32+v_bitop3_b32+2*v_fma_f32"
    if costs and mixk in costs:
        pair_alone = costs["v_mad_u64_u32"] + costs["v_bitop3_b32"]
        pair_ctx = 4 * costs[mixk] - 2 * costs["v_fma_f32"]
        n_pairs = min(ops.get("v_mad_u64_u32", 0), ops.get("v_bitop3_b32", 0))
        busy_ctx = busy - n_pairs * (pair_alone - pair_ctx)
    if costs:
        print(f"\n{'opcode':24s} {'count':>6s} {'ns each':>8s} {'ns':>9s} {'share':>6s}")
        for o, c, ns in rows:
            print(f"{o:24s} {c:6d} {ns:8.3f} {c * ns:9.1f} {c * ns / busy:6.1%}")
        print(f"busy SIMD time per wave-step (sum of VALU count x measured issue cost): {busy:.1f} ns"
              + (f"; unpriced opcodes priced as v_fma_f32: {unpriced}" if unpriced else ""))
        if busy_ctx is not None:
            print(f"the same with the Philox pairs at their in-context price ({n_pairs} x ({pair_ctx:.3f} instead of {pair_alone:.3f}) ns): "
                  f"{busy_ctx:.1f} ns")
        if a.ns_per_wave_step:
            print(f"measured SIMD time per wave-step: {a.ns_per_wave_step:.1f} ns  ->  sum of stand-alone costs / measured = "
                  f"{busy / a.ns_per_wave_step:.3f}" + (f"; with in-context Philox price = {busy_ctx / a.ns_per_wave_step:.3f} "
                  "(the cost-weighted issue fraction: how much of the step's SIMD time its own instructions account for)"
                  if busy_ctx is not None else ""))
    if a.json:
        json.dump({"kernel": name, "path": path, "valu_on_path": valu, "salu_on_path": salu, "histogram": dict(ops),
                   "busy_ns_per_wave_step": busy if costs else None, "unpriced": unpriced,
                   "pmc_check": {"valu_per_wave_step": a.valu_per_wave_step, "salu_per_wave_step": a.salu_per_wave_step, "ok": ok},
                   "ns_per_wave_step": a.ns_per_wave_step,
                   "sum_of_standalone_costs_over_measured": (busy / a.ns_per_wave_step) if (costs and a.ns_per_wave_step) else None,
                   "busy_ns_per_wave_step_in_context": busy_ctx,
                   "issue_cost_weighted": ((busy_ctx if busy_ctx is not None else busy) / a.ns_per_wave_step)
                   if (costs and a.ns_per_wave_step) else None,
                   "costs": os.path.relpath(a.costs, ROOT), "waves": a.waves, "lib_sha256": lib_sha}, open(a.json, "w"), indent=1)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
