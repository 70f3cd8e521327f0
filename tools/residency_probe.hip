// Where do the waves of a launch really run, how many share a SIMD at a time, and how fast does that SIMD issue?
// Ground truth for tools/issue_cost.hip's "N waves per SIMD" (round 4: VERDICT r03 #2).  Every wave of a grid of
// 256-thread blocks runs a loop of ITERS x 64 independent-accumulator VALU instructions of one kind and records
//   HW_REG_HW_ID (SIMD, CU, SE), HW_REG_XCC_ID, s_memrealtime at start and end (one 100 MHz clock for the whole chip),
//   and the s_memtime (shader clock) it took.
// The host sorts the waves by (XCC, SE, CU, SIMD), counts how many lifetimes overlap on each SIMD (time-weighted), and
// reports, per blocks-per-CU setting: the distribution of waves per SIMD, the mean overlap, cycles per wave-instruction of
// ONE wave, and SIMD cycles per wave-instruction = that / overlap.
//   hipcc --offload-arch=gfx950 -O3 tools/residency_probe.hip -o tools/residency_probe && tools/residency_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define ITERS 4000

struct Rec {
  unsigned hw_id, xcc_id;
  unsigned long long r0, r1, cyc;
};

enum { OP_ADD_ACC, OP_ADD_INDEP, OP_FMA_ACC, OP_MAX3, OP_EXP, OP_MAD64, OP_MIX, N_OP };
static const char *kOpNames[N_OP] = {"v_add_f32 v,v,c (accumulate)", "v_add_f32 v,c1,c2 (independent)", "v_fma_f32 v,v,c1,c2",
                                     "v_max3_f32 v,v,c1,c2", "v_exp_f32 v,v", "v_mad_u64_u32",
                                     "mix: 2 fma + add + max3 + mad64 + bitop3 + exp/8 (a step-like blend)"};

template <int OP>
__global__ void __launch_bounds__(256) k(Rec *rec, unsigned seed, unsigned *sink) {
  float f[8];
  unsigned a[8];
  unsigned long long p[8];
  const float c1 = 1.0001f + seed * 1e-9f, c2 = 0.5f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + threadIdx.x * 8 + i;
    f[i] = 0.5f + (float)(a[i] & 1023) * 1e-4f;
    p[i] = a[i];
  }
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == OP_ADD_ACC) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c2));
        if (OP == OP_ADD_INDEP) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[i]) : "v"(c1), "v"(c2));
        if (OP == OP_FMA_ACC) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c1), "v"(c2));
        if (OP == OP_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c1), "v"(c2));
        if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
        if (OP == OP_MAD64) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p[i]) : "s"(0xD2511F53u), "v"(a[i]) : "vcc");
          a[i] = (unsigned)p[i];
        }
        if (OP == OP_MIX) {  // eight instructions per slot, one slot in eight carries the transcendental
          if (i == 0) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[r]) : "v"(c1), "v"(c2));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(r + 1) & 7]) : "v"(c1), "v"(c2));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(r + 2) & 7]) : "v"(c2));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[(r + 3) & 7]) : "v"(c1), "v"(c2));
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p[r]) : "s"(0xD2511F53u), "v"(a[r]) : "vcc");
            asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[(r + 1) & 7]) : "v"((unsigned)p[r]), "s"(seed));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[(r + 4) & 7]) : "v"(c1));
            if (r == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(f[5]));
            else asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(r + 5) & 7]) : "v"(c2));
          }
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    Rec rr;
    rr.hw_id = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID
    rr.xcc_id = __builtin_amdgcn_s_getreg(20 | (31 << 11));  // HW_REG_XCC_ID
    rr.r0 = r0, rr.r1 = r1, rr.cyc = t1 - t0;
    rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = rr;
  }
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + (unsigned)f[i] + (unsigned)p[i];
  if (s == 0x12345678u) *sink = s;
}

template <int OP>
static void run(Rec *drec, unsigned *sink, int cus, int blocks_per_cu, bool first) {
  const int blocks = cus * blocks_per_cu, waves = blocks * 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) k<OP><<<blocks, 256>>>(drec, 1, sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(drec, 1, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<Rec> rec(waves);
  (void)hipMemcpy(rec.data(), drec, waves * sizeof(Rec), hipMemcpyDeviceToHost);
  // per SIMD: the waves that ran there and the time-weighted number alive at once
  std::map<unsigned long long, std::vector<const Rec *>> by_simd;
  for (const Rec &r : rec) {
    const unsigned simd = (r.hw_id >> 4) & 3, cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7;
    const unsigned long long key = ((unsigned long long)(r.xcc_id & 15) << 24) | (se << 16) | (sh << 12) | (cu << 4) | simd;
    by_simd[key].push_back(&r);
  }
  std::map<int, int> hist;  // waves per SIMD -> number of SIMDs
  double overlap_sum = 0.0, life_sum = 0.0;
  std::vector<double> cyc, ghz;
  for (auto &kv : by_simd) {
    hist[(int)kv.second.size()]++;
    // time-weighted concurrency seen by the waves of this SIMD: sum over pairs of overlap / own lifetime
    for (const Rec *a : kv.second) {
      double ov = 0.0;
      for (const Rec *b : kv.second) {
        const double lo = (double)std::max(a->r0, b->r0), hi = (double)std::min(a->r1, b->r1);
        if (hi > lo) ov += hi - lo;
      }
      overlap_sum += ov;
      life_sum += (double)(a->r1 - a->r0);
    }
  }
  for (const Rec &r : rec) {
    cyc.push_back((double)r.cyc);
    ghz.push_back((double)r.cyc / (double)(r.r1 - r.r0) * 0.1);
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(ghz.begin(), ghz.end());
  const double n_inst = (double)ITERS * 64 * (OP == OP_MIX ? 1.0 : 1.0);
  const double conc = overlap_sum / life_sum;              // mean number of waves alive on a wave's SIMD during its life (itself included)
  const double own = cyc[waves / 2] / n_inst;              // cycles per instruction of one wave's stream
  printf("%s    {\"blocks_per_cu\": %d, \"simds_used\": %zu, \"waves_per_simd_histogram\": {", first ? "" : ",\n", blocks_per_cu, by_simd.size());
  bool f1 = true;
  for (auto &h : hist) {
    printf("%s\"%d\": %d", f1 ? "" : ", ", h.first, h.second);
    f1 = false;
  }
  printf("}, \"mean_waves_alive_per_simd\": %.3f, \"wave_cycles_per_instruction\": %.3f, \"simd_cycles_per_wave_instruction\": %.3f, "
         "\"clock_ghz\": %.3f, \"kernel_ms\": %.4f, \"ns_per_wave_instruction_if_evenly_spread\": %.4f}",
         conc, own, own / conc, ghz[waves / 2], ms, ms * 1e6 / (n_inst * blocks_per_cu));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
}

template <int OP>
static void sweep(Rec *drec, unsigned *sink, int cus) {
  printf("%s  \"%s\": [\n", OP ? ",\n" : "", kOpNames[OP]);
  const int bpc[] = {1, 2, 3, 4, 6, 8};
  for (int i = 0; i < 6; ++i) run<OP>(drec, sink, cus, bpc[i], i == 0);
  printf("\n  ]");
  if constexpr (OP + 1 < N_OP) sweep<OP + 1>(drec, sink, cus);
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  Rec *drec;
  unsigned *sink;
  (void)hipMalloc(&drec, (size_t)cus * 8 * 4 * sizeof(Rec));
  (void)hipMalloc(&sink, 64);
  printf("{\n  \"device\": \"%s\", \"compute_units\": %d, \"iters\": %d,\n", prop.gcnArchName, cus, ITERS);
  sweep<0>(drec, sink, cus);
  printf("\n}\n");
  return 0;
}
