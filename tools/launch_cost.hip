// Host cost of ptrwm_run for one-step launches, straight through the C ABI (development aid).
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/launch_cost.hip -o tools/launch_cost -ldl && tools/launch_cost <lib.so>
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ptrwm.h"

__global__ void tiny(float *p) { p[threadIdx.x] += 1.0f; }

int main(int argc, char **argv) {
  void *h = dlopen(argv[1], RTLD_NOW);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
  auto run = (decltype(&ptrwm_run))dlsym(h, "ptrwm_run");
  const int D = 30, T = 8;
  const long long C = 64;
  float *state, *logp, *beta, *ts;
  hipMalloc(&state, C * T * D * 4); hipMemset(state, 0, C * T * D * 4);
  hipMalloc(&logp, C * T * 4); hipMemset(logp, 0, C * T * 4);
  std::vector<float> b(T), s(T);
  for (int t = 0; t < T; ++t) { b[t] = std::pow(0.01f, t / float(T - 1)); s[t] = std::sqrt(2.38f * 2.38f / D / b[t]); }
  hipMalloc(&beta, T * 4); hipMemcpy(beta, b.data(), T * 4, hipMemcpyHostToDevice);
  hipMalloc(&ts, T * 4); hipMemcpy(ts, s.data(), T * 4, hipMemcpyHostToDevice);
  ptrwm_target_desc td; memset(&td, 0, sizeof td);
  td.kind = PTRWM_TARGET_ROUGH_CARPET; td.dim = D;
  td.p[0] = -15; td.p[1] = 0; td.p[2] = 15; td.p[3] = std::log(0.5f); td.p[4] = std::log(0.3f); td.p[5] = std::log(0.2f);
  ptrwm_proposal_desc pd; memset(&pd, 0, sizeof pd);
  pd.kind = PTRWM_PROPOSAL_NORMAL; pd.temp_scale = ts;
  ptrwm_run_args a; memset(&a, 0, sizeof a);
  a.struct_size = sizeof a; a.n_temps = T; a.n_chains = C; a.state = state; a.logp = logp; a.beta = beta;
  a.n_steps = 1; a.swap_every = 10; a.seed = 1;
  hipStream_t st; hipStreamCreate(&st);
  for (int which = 0; which < 3; ++which) {
    hipStream_t s0 = which == 0 ? nullptr : st;
    for (int i = 0; i < 100; ++i) { a.step0 = i; if (which < 2) run(&td, &pd, &a, s0); else hipLaunchKernelGGL(tiny, 1, 64, 0, st, state); }
    hipDeviceSynchronize();
    const int n = 2000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
      a.step0 = 100 + i;
      if (which < 2) { int rc = run(&td, &pd, &a, s0); if (rc) { printf("rc %d\n", rc); return 1; } }
      else hipLaunchKernelGGL(tiny, 1, 64, 0, st, state);
    }
    auto t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("%s: issue %.1f us/launch, end-to-end %.1f us/launch\n",
           which == 0 ? "ptrwm_run null stream" : which == 1 ? "ptrwm_run created stream" : "tiny kernel created stream",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
  }
  // (a) pure issue cost: a few launches into an empty queue
  auto logd = (decltype(&ptrwm_logdensity))dlsym(h, "ptrwm_logdensity");
  for (int rep = 0; rep < 3; ++rep) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 4; ++i) { a.step0 = 5000 + i; run(&td, &pd, &a, st); }
    auto t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("4 launches into an empty queue: issue %.1f us each, until done %.1f us total\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / 4,
           std::chrono::duration<double, std::micro>(t2 - t0).count());
  }
  // (b) another kernel of the same library
  {
    const int n = 2000;
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) logd(&td, state, logp, C * T, st);
    auto t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("ptrwm_logdensity: issue %.1f us/launch, end-to-end %.1f us/launch\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
  }
  // (c) many steps per launch: the per-launch cost amortised
  for (int ns : {1, 10, 100}) {
    a.n_steps = ns;
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; ++i) { a.step0 = 10000 + i * ns; run(&td, &pd, &a, st); }
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("n_steps=%d: %.1f us/launch end-to-end\n", ns, std::chrono::duration<double, std::micro>(t2 - t0).count() / 200);
  }
  return 0;
}
