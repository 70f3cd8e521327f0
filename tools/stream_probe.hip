// Development probe for the streaming form of the fused step kernel (kernel.h, STREAM): BASELINE configs[2]'s shape
// (RoughCarpet dim 30 modes +-15, Normal proposal, 32 geometric temperatures, swap_every 10) driven WITHOUT the C ABI, the
// classic and the streaming kernel side by side from the same initial state: bitwise comparison of everything they write,
// then time per launch of each (HIP events around a train of launches).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-mllvm ... as csrc/Makefile SCHED] tools/stream_probe.hip -o tools/stream_probe
//   tools/stream_probe [chains=65536] [n_steps=1] [wg_per_cu=3] [launches=200]
#include "../rwm-pt-pytorch_amd/csrc/kernel.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace ptrwm;

#ifndef PROBE_DP
#define PROBE_DP 30
#endif
constexpr int DP = PROBE_DP;
typedef RoughCarpet2<DP> Tgt;
typedef NormalProposal<DP> Prop;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

struct Bufs {
  float *state, *logp;
  long long *n_accept, *swap_accept, *last_ord;
  double *sq_jump;
};

static Bufs alloc(long long reps, int D) {
  Bufs b;
  CK(hipMalloc(&b.state, reps * D * 4));
  CK(hipMalloc(&b.logp, reps * 4));
  CK(hipMalloc(&b.n_accept, reps * 8));
  CK(hipMalloc(&b.swap_accept, reps * 8));
  CK(hipMalloc(&b.last_ord, reps * 8));
  CK(hipMalloc(&b.sq_jump, reps * 8));
  return b;
}

static void reset(const Bufs &b, long long reps, int D, const TParams &tp) {
  CK(hipMemset(b.state, 0, reps * D * 4));
  CK(hipMemset(b.n_accept, 0, reps * 8));
  CK(hipMemset(b.swap_accept, 0, reps * 8));
  CK(hipMemset(b.last_ord, 0, reps * 8));
  CK(hipMemset(b.sq_jump, 0, reps * 8));
  hipLaunchKernelGGL((ptrwm_logdensity_kernel<RoughCarpet<DP>, DP>), dim3((unsigned)((reps + 255) / 256)), dim3(256), 0, 0, b.state,
                     b.logp, reps, D, tp);
  CK(hipDeviceSynchronize());
}

template <bool STREAM>
static void launch(KArgs k, const Bufs &b, long long step0, int n_steps, int se, unsigned grid) {
  k.state = b.state;
  k.logp = b.logp;
  k.n_accept = b.n_accept;
  k.sq_jump = b.sq_jump;
  k.swap_accept = b.swap_accept;
  k.last_swap_ordinal = b.last_ord;
  k.step0 = step0;
  k.n_steps = n_steps;
  k.burn_left = 0;
  k.first_swap_event = step0 / se;
  k.steps_to_swap = (int)(se - step0 % se);
  const unsigned lds = step_kernel_lds_bytes(kBlockThreads, DP, STREAM);
  hipLaunchKernelGGL((ptrwm_step_kernel<Tgt, Prop, DP, true, false, STREAM>), dim3(grid), dim3(kBlockThreads), lds, 0, k);
}

// reference: a plain float4 copy of the state array (read + write of the same bytes a launch moves for the state)
__global__ void __launch_bounds__(256) copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

template <class T>
static long long diff(const T *a, const T *b, long long n) {
  std::vector<T> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * sizeof(T), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * sizeof(T), hipMemcpyDeviceToHost));
  long long bad = 0;
  for (long long i = 0; i < n; ++i) bad += memcmp(&ha[i], &hb[i], sizeof(T)) != 0;
  return bad;
}

int main(int argc, char **argv) {
  const long long C = argc > 1 ? atoll(argv[1]) : 65536;
  const int n_steps = argc > 2 ? atoi(argv[2]) : 1;
  const int wg_per_cu = argc > 3 ? atoi(argv[3]) : stream_waves_per_simd(DP);
  const int launches = argc > 4 ? atoi(argv[4]) : 200;
  const int T = 32, D = DP, se = 10;
  const long long reps = C * T;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;

  TParams tp;
  memset(&tp, 0, sizeof(tp));
  tp.p[0] = -15.0f, tp.p[1] = 0.0f, tp.p[2] = 15.0f;
  tp.p[3] = logf(0.5f), tp.p[4] = logf(0.3f), tp.p[5] = logf(0.2f);
  tp.p[6] = 0.0f;
  tp.p[7] = -(float)D * 0.91893853320467274178f;
  std::vector<float> hb(T), hs(T);
  for (int t = 0; t < T; ++t) {
    hb[t] = (float)pow(0.01, (double)t / (T - 1));
    hs[t] = (float)sqrt(2.38 * 2.38 / D / (double)hb[t]);
  }
  float *beta, *tscale;
  CK(hipMalloc(&beta, T * 4));
  CK(hipMalloc(&tscale, T * 4));
  CK(hipMemcpy(beta, hb.data(), T * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(tscale, hs.data(), T * 4, hipMemcpyHostToDevice));

  KArgs k;
  memset(&k, 0, sizeof(k));
  k.beta = beta;
  k.temp_scale = tscale;
  k.n_chains = C;
  k.chain_offset = 0;
  k.n_temps = T;
  k.dim = D;
  k.swap_every = se;
  k.swap_mode = PTRWM_SWAP_EXCHANGE;
  k.swap_order = PTRWM_ORDER_SEQUENTIAL;
  k.chains_per_wave = 64 / T;
  k.k0 = 42, k.k1 = 0;
  k.tp = tp;
  k.pp.dim_scale = nullptr;
  k.pp.inv_dim = 1.0f / D;

  const long long n_groups = C / k.chains_per_wave;
  const unsigned grid_classic = (unsigned)((n_groups + kWavesPerBlock - 1) / kWavesPerBlock);
  // streaming grid: as many workgroups as the device holds at wg_per_cu per CU, trimmed so that every wave walks the
  // same number of groups (+-1)
  const long long max_waves = (long long)cus * wg_per_cu * kWavesPerBlock;
  const long long rounds = (n_groups + max_waves - 1) / max_waves;
  const long long waves = (n_groups + rounds - 1) / rounds;
  const unsigned grid_stream = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);

  const unsigned lds = step_kernel_lds_bytes(kBlockThreads, DP), lds_s = step_kernel_lds_bytes(kBlockThreads, DP, true);
  CK(hipFuncSetAttribute((const void *)ptrwm_step_kernel<Tgt, Prop, DP, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute((const void *)ptrwm_step_kernel<Tgt, Prop, DP, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_s));
  int occ_c = 0, occ_s = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_c, ptrwm_step_kernel<Tgt, Prop, DP, true, false, false>, kBlockThreads, lds));
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_s, ptrwm_step_kernel<Tgt, Prop, DP, true, false, true>, kBlockThreads, lds_s));

  Bufs A = alloc(reps, D), B = alloc(reps, D);
  reset(A, reps, D, tp);
  reset(B, reps, D, tp);
  // parity: 25 launches each (two and a half swap periods at one step per launch)
  for (int i = 0; i < 25; ++i) {
    launch<false>(k, A, (long long)i * n_steps, n_steps, se, grid_classic);
    launch<true>(k, B, (long long)i * n_steps, n_steps, se, grid_stream);
  }
  CK(hipDeviceSynchronize());
  const long long d_state = diff(A.state, B.state, reps * D), d_lp = diff(A.logp, B.logp, reps),
                  d_acc = diff(A.n_accept, B.n_accept, reps), d_sq = diff(A.sq_jump, B.sq_jump, reps),
                  d_sw = diff(A.swap_accept, B.swap_accept, reps), d_lo = diff(A.last_ord, B.last_ord, reps);
  // something must have happened
  std::vector<long long> hacc(reps);
  CK(hipMemcpy(hacc.data(), A.n_accept, reps * 8, hipMemcpyDeviceToHost));
  long long tot_acc = 0;
  for (long long i = 0; i < reps; ++i) tot_acc += hacc[i];

  auto time_it = [&](bool stream, const Bufs &b) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    long long s = 25LL * n_steps;
    for (int i = 0; i < 20; ++i, s += n_steps)
      stream ? launch<true>(k, b, s, n_steps, se, grid_stream) : launch<false>(k, b, s, n_steps, se, grid_classic);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < launches; ++i, s += n_steps)
      stream ? launch<true>(k, b, s, n_steps, se, grid_stream) : launch<false>(k, b, s, n_steps, se, grid_classic);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms / launches;
  };
  double tc[3], ts[3];
  for (int r = 0; r < 3; ++r) {
    tc[r] = time_it(false, A);
    ts[r] = time_it(true, B);
  }
  // the copy reference, same bytes as the state traffic of one launch
  double tcopy[3];
  {
    const long long n4 = reps * D / 4;
    for (int r = 0; r < 3; ++r) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(copy_kernel, dim3(cus * 8), dim3(256), 0, 0, (const float4 *)A.state, (float4 *)B.state, n4);
      CK(hipEventRecord(e0));
      for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(copy_kernel, dim3(cus * 8), dim3(256), 0, 0, (const float4 *)A.state, (float4 *)B.state, n4);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      tcopy[r] = (double)ms / 50;
    }
  }
  const double state_bytes = 2.0 * 4.0 * D * reps;
  const double alg_bytes = (8.0 * D + 24.0) * reps;
  printf("{\"chains\": %lld, \"temps\": %d, \"dim\": %d, \"n_steps\": %d, \"cus\": %d, \"wg_per_cu\": %d, \"grid_classic\": %u, "
         "\"grid_stream\": %u, \"rounds\": %lld, \"occupancy_classic\": %d, \"occupancy_stream\": %d, "
         "\"mismatch\": {\"state\": %lld, \"logp\": %lld, \"n_accept\": %lld, \"sq_jump\": %lld, \"swap_accept\": %lld, \"last_ord\": %lld}, "
         "\"accepted_total\": %lld, \"classic_ms\": [%.5f, %.5f, %.5f], \"stream_ms\": [%.5f, %.5f, %.5f], "
         "\"classic_frac_264B\": %.4f, \"stream_frac_264B\": %.4f, \"copy_ms\": %.5f, \"copy_TBps\": %.3f, "
         "\"classic_state_TBps\": %.3f, \"stream_state_TBps\": %.3f}\n",
         C, T, D, n_steps, cus, wg_per_cu, grid_classic, grid_stream, rounds, occ_c, occ_s, d_state, d_lp, d_acc, d_sq, d_sw, d_lo,
         tot_acc, tc[0], tc[1], tc[2], ts[0], ts[1], ts[2], alg_bytes / (tc[1] * 1e-3) / 8e12, alg_bytes / (ts[1] * 1e-3) / 8e12, tcopy[1], state_bytes / (tcopy[1] * 1e-3) / 1e12,
         state_bytes / (tc[1] * 1e-3) / 1e12, state_bytes / (ts[1] * 1e-3) / 1e12);
  return (d_state | d_lp | d_acc | d_sq | d_sw | d_lo) != 0;
}
