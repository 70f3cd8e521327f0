"""The issue ceiling of the headline kernel's OWN instruction mix, measured (round 4: VERDICT r03 "next round" #2).

    python tools/mix_probe.py gen  profiles/r04_issue_model.json          -> tools/mix_probe.hip, tools/mix_probe (hipcc)
    tools/mix_probe > gpurun_out/r04_mix_probe.json                        (on the GPU box)
    python tools/mix_probe.py read gpurun_out/r04_mix_probe.json profiles/r04_issue_model.json   -> adds "measured_mix" to the model

tools/issue_model.py prices one Metropolis step by adding up what each opcode costs when a SIMD issues nothing else
(tools/issue_cost.hip).  That sum is NOT a floor: different kinds of "slow" instruction (64-bit multiplies, SGPR operands,
three-operand selects, transcendentals) are slow for different reasons and overlap when they are mixed, as the step mixes them -
the headline kernel runs 7 % FASTER than the sum.  This probe measures the mix itself: a kernel whose loop body is the
step path's VALU and SALU instructions - same opcodes, same operand kinds (VGPR / SGPR / literal), same count - with every
DEPENDENCY removed (sources are registers nothing writes, destinations rotate over a pool nothing reads), run at the headline
kernel's own residency (128 VGPRs: four waves per SIMD, 256-thread workgroups, 8 192 of them).  Two orders: the kernel's own
program order, and every opcode spread evenly through the body.  No schedule of these instructions can issue faster than the
better of the two on this hardware; how close the real step - with its dependencies, LDS and memory waits - comes to it is
`roofline.frac_of_measured_mix` in the bench line.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CONST, CONST0 = 16, 96  # v96..v111 are never written inside the loop
N_DEST = 48               # v0..v47 are never read
ITERS = 400


def template(op, cls, k):
    """One instruction without dependencies: destination from the rotating pool, sources from the constants."""
    d = f"v{k % N_DEST}"
    dp = f"v[{2 * (k % (N_DEST // 2))}:{2 * (k % (N_DEST // 2)) + 1}]"
    a, b, c = (f"v{CONST0 + (k + o) % N_CONST}" for o in (0, 5, 11))  # three different register banks
    sg = cls == "full+sgpr"
    if cls == "salu":
        return "s_add_i32 s46, s46, s40"
    if op == "v_mad_u64_u32":
        return f"v_mad_u64_u32 {dp}, s[42:43], {a}, s40, 0"  # (the Philox multipliers are SGPR constants in the kernel)
    if op in ("v_mul_hi_u32", "v_mul_lo_u32"):
        return f"{op} {d}, {a}, {b}"
    if op == "v_bitop3_b32":
        return f"v_bitop3_b32 {d}, {a}, {b}, {'s41' if sg else c} bitop3:0x96"
    if op == "v_and_or_b32":
        return f"v_and_or_b32 {d}, {a}, s40, {b}"
    if op in ("v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_min_f32",
              "v_max_f32", "v_subrev_f32", "v_subrev_u32"):
        return f"{op} {d}, {'s40' if sg else a}, {b}"
    if op in ("v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32"):
        return f"{op} {d}, 9, {a}"
    if op in ("v_fma_f32", "v_max3_f32", "v_med3_f32", "v_min3_f32", "v_mad_u32_u24", "v_lshl_add_u32", "v_add3_u32"):
        return f"{op} {d}, {a}, {'s40' if sg else b}, {c}"
    if op == "v_fmamk_f32":
        return f"v_fmamk_f32 {d}, {a}, 0x3f317218, {b}"
    if op == "v_fmaak_f32":
        return f"v_fmaak_f32 {d}, {a}, {b}, 0x3f317218"
    if op == "v_fmac_f32":
        return f"v_fmac_f32 {d}, {a}, {b}"
    if op == "v_mov_b32":
        return f"v_mov_b32 {d}, {'s40' if sg else a}"
    if op in ("v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_cvt_f32_u32", "v_cvt_f32_i32",
              "v_cvt_u32_f32", "v_cvt_f64_f32", "v_rndne_f32", "v_fract_f32"):
        return f"{op} {dp if op == 'v_cvt_f64_f32' else d}, {a}"
    if op == "v_cndmask_b32":
        return f"v_cndmask_b32 {d}, {a}, {b}, s[44:45]"  # (the step's selects take their mask from an SGPR pair)
    if op.startswith("v_cmp_"):
        return f"{op} s[42:43], {a}, {b}"
    if op == "v_add_f64":
        return f"v_add_f64 {dp}, v[{CONST0}:{CONST0 + 1}], v[{CONST0 + 2}:{CONST0 + 3}]"
    if op in ("v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"):
        return f"{op} {d}, s40, {a}"
    if op in ("v_lshl_add_u64",):
        return f"{op} {dp}, v[{CONST0}:{CONST0 + 1}], 2, v[{CONST0 + 2}:{CONST0 + 3}]"
    raise SystemExit(f"mix_probe: no template for {op} [{cls}]")


def rewrite(op, cls, args, k, recent=0):
    """The instruction as the kernel has it - same encoding, operand kinds, literals and modifiers - with every register
    renamed so that nothing depends on anything: VGPR sources -> the constant pool, VGPR destinations -> the rotating pool,
    SGPR sources -> s40 / s[44:45], SGPR destinations -> s[42:43]; vcc stays vcc.  None if the shape is not understood
    (the caller falls back to the class template)."""
    import re
    if cls == "salu" or "dpp" in op or "sdwa" in op:
        return None
    m = re.search(r"\s+(bitop3:|clamp|op_sel|mul:|div:|row_|quad_perm|neg_|bound_ctrl|bank_mask)", args)  # modifiers follow the operands
    head, tail = (args[:m.start()], args[m.start():].strip()) if m else (args, "")
    ops = [x.strip() for x in re.split(r",(?![^\[]*\])", head)]
    if not ops or not ops[0]:
        return None
    n_dst = 2 if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_div_scale")) else 1
    src_regs = [f"v{CONST0 + (k + o) % N_CONST}" for o in (0, 5, 11, 14)]
    out, si = [], 0
    for i, x in enumerate(ops):
        m = re.match(r"^(-?\|?)(.*?)(\|?)$", x)
        pre, core, post = m.group(1), m.group(2), m.group(3)
        if i < n_dst:
            if re.match(r"^v\d+$", core):
                core = f"v{(k + 2) % N_DEST}"
            elif re.match(r"^v\[(\d+):(\d+)\]$", core):
                a, b = map(int, re.match(r"^v\[(\d+):(\d+)\]$", core).groups())
                if b - a != 1:
                    return None
                core = f"v[{2 * (k % (N_DEST // 2))}:{2 * (k % (N_DEST // 2)) + 1}]"
            elif re.match(r"^s\[\d+:\d+\]$", core):
                core = "s[42:43]"
            elif core != "vcc":
                return None
        else:
            if re.match(r"^v\d+$", core):
                # (recent > 0: the first VGPR source is the result of the instruction `recent` places earlier - a dependency
                # far enough back never to stall with four waves resident, but one the operand may arrive by without a
                # register-file read, as most operands of the real step do)
                core = f"v{(k - recent + 2) % N_DEST}" if (recent and si == 0) else src_regs[si % 4]
                si += 1
            elif re.match(r"^v\[(\d+):(\d+)\]$", core):
                a, b = map(int, re.match(r"^v\[(\d+):(\d+)\]$", core).groups())
                if b - a != 1:
                    return None
                core = f"v[{CONST0 + 2 * (si % 4)}:{CONST0 + 2 * (si % 4) + 1}]"
                si += 1
            elif re.match(r"^s\d+$", core):
                core = "s40"
            elif re.match(r"^s\[\d+:\d+\]$", core):
                core = "s[44:45]"
            elif core in ("vcc", "vcc_lo", "vcc_hi", "exec", "exec_lo", "exec_hi") or re.match(r"^(-?\d+(\.\d+)?|0x[0-9a-fA-F]+)$", core):
                pass
            else:
                return None
        out.append(pre + core + post)
    return f"{op} " + ", ".join(out) + (" " + tail if tail else "")


def spread(seq):
    """The same multiset with every (opcode, class) spread evenly through the body (largest-remainder interleave)."""
    import collections
    groups = collections.defaultdict(list)  # (opcode, class) -> its instructions, each with its own operand text
    for e in seq:
        groups[(e[0], e[1])].append(e)
    n = len(seq)
    slots = []
    for key, es in groups.items():
        slots += [((i + 0.5) * n / len(es), i, key) for i in range(len(es))]
    slots.sort()
    return [groups[key][i] for _, i, key in slots]


N_FALLBACK = [0]


def body(seq, recent=0):
    lines = []
    for k, e in enumerate(seq):
        o, c = e[0], e[1]
        t = rewrite(o, c, e[2], k, recent) if len(e) > 2 else None
        if t is None:
            t = template(o, c, k)
            N_FALLBACK[0] += c != "salu"
        lines.append(f'      "{t}\\n"')
    return "\n".join(lines)


def gen(model_json):
    m = json.load(open(model_json))
    seq = m["sequence"]
    nv = sum(1 for e in seq if e[1] != "salu")
    clob = ", ".join(f'"v{i}"' for i in range(122)) + ', "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s48", "vcc", "scc"'
    init = "\n".join(f'      "v_mov_b32 v{CONST0 + i}, {0.3 + 0.01 * i:.2f}\\n"' for i in range(N_CONST))

    def kernel(name, s, recent=0):
        return f"""__global__ void __launch_bounds__(256) {name}(unsigned long long *cyc, float *sink) {{
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  asm volatile(
{init}
      "s_mov_b32 s40, 0xD2511F53\\n s_mov_b32 s41, 0x9E3779B9\\n s_mov_b64 s[44:45], 0x5555\\n s_mov_b32 s46, 0\\n"
      "s_movk_i32 s48, {ITERS}\\n"
      "1:\\n"
{body(s, recent)}
      "s_sub_i32 s48, s48, 1\\n s_cmp_lg_u32 s48, 0\\n s_cbranch_scc1 1b\\n"
      "v_mov_b32 v121, v0\\n"  // (122 VGPRs clobbered + the compiler's own: 128 allocated - the headline kernel's residency, four waves per SIMD)
      ::: {clob});
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {{
    cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;      // shader clock
    cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;  // 100 MHz
  }}
  if (t1 == 0x12345) *sink = 1.0f;
}}
"""
    src = f"""// GENERATED by tools/mix_probe.py gen {os.path.relpath(model_json, ROOT)} - do not edit.  {nv} VALU + {len(seq) - nv} SALU instructions per iteration:
// the step path of {m['kernel'][:90]}... without its dependencies.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
{kernel('mix_in_program_order', seq)}
{kernel('mix_spread_evenly', spread(seq))}
{kernel('mix_in_program_order_recent_sources', seq, 6)}
template <class K>
static void run(K k, const char *name, unsigned long long *dc, float *sink, bool last) {{
  const int blocks = 8192, waves = blocks * 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<<<blocks, 256>>>(dc, sink);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {{
    (void)hipEventRecord(e0);
    k<<<blocks, 256>>>(dc, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }}
  std::vector<unsigned long long> both(2 * waves), c(waves);
  (void)hipMemcpy(both.data(), dc, 2 * waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> ghz(waves);
  for (int i = 0; i < waves; ++i) {{
    c[i] = both[2 * i];
    ghz[i] = (double)both[2 * i] / ((double)both[2 * i + 1] * 10.0);
  }}
  std::sort(c.begin(), c.end());
  std::sort(ghz.begin(), ghz.end());
  // SIMD time per wave-iteration, as tools/issue_model.py defines it for the real kernel: kernel time x 1024 SIMDs / (waves x iterations)
  const double ns = (double)best * 1e6 * 1024.0 / ((double)waves * {ITERS});
  printf("  \\"%s\\": {{\\"kernel_ms\\": %.4f, \\"ns_of_simd_time_per_wave_iteration\\": %.2f, \\"shader_clock_ghz\\": %.4f, "
         "\\"cycles_of_simd_time_per_wave_iteration\\": %.1f, \\"median_wave_lifetime_cycles_per_iteration\\": %.1f}}%s\\n",
         name, best, ns, ghz[waves / 2], ns * ghz[waves / 2], (double)c[waves / 2] / {ITERS}, last ? "" : ",");
}}
int main() {{
  unsigned long long *dc;
  float *sink;
  (void)hipMalloc(&dc, 2 * 8192 * 4 * sizeof(unsigned long long));
  (void)hipMalloc(&sink, 64);
  printf("{{\\n  \\"valu_per_iteration\\": {nv}, \\"salu_per_iteration\\": {len(seq) - nv}, \\"iterations\\": {ITERS}, \\"workgroups\\": 8192, \\"threads\\": 256,\\n");
  run(mix_in_program_order, "program_order", dc, sink, false);
  run(mix_spread_evenly, "spread_evenly", dc, sink, false);
  run(mix_in_program_order_recent_sources, "program_order_first_source_is_a_recent_result", dc, sink, true);
  printf("}}\\n");
  return 0;
}}
"""
    out = os.path.join(ROOT, "tools", "mix_probe.hip")
    open(out, "w").write(src)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", out, "-o", os.path.join(ROOT, "tools", "mix_probe")])
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", os.path.join(ROOT, "tools", "mix_probe")], capture_output=True, text=True)
    print(f"tools/mix_probe.hip: {nv} VALU + {len(seq) - nv} SALU per iteration ({N_FALLBACK[0] // 1} VALU instructions of the two bodies from class "
          "templates, the rest with the kernel's own operand forms); built tools/mix_probe")


def read(probe_json, model_json):
    p, m = json.load(open(probe_json)), json.load(open(model_json))
    best = min(p["program_order"]["ns_of_simd_time_per_wave_iteration"], p["spread_evenly"]["ns_of_simd_time_per_wave_iteration"])
    m["measured_mix"] = {"ns_of_simd_time_per_wave_step": best, "program_order_ns": p["program_order"]["ns_of_simd_time_per_wave_iteration"],
                         "spread_evenly_ns": p["spread_evenly"]["ns_of_simd_time_per_wave_iteration"],
                         "valu_per_iteration": p["valu_per_iteration"], "source": os.path.relpath(probe_json, ROOT)}
    if m.get("ns_per_wave_step"):
        m["measured_mix"]["kernel_ns_per_wave_step"] = m["ns_per_wave_step"]
        m["measured_mix"]["frac_of_measured_mix"] = best / m["ns_per_wave_step"]
    json.dump(m, open(model_json, "w"), indent=1)
    print(json.dumps(m["measured_mix"], indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "gen":
        gen(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "read":
        read(sys.argv[2], sys.argv[3])
    else:
        raise SystemExit(__doc__)
