/*
 * ptrwm_oracle.c -- CPU restatement of the reference's PT-RWM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or
 * executed by the product path (rwm-pt-pytorch_amd/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity pin: this restatement is checked against golden vectors captured by
 * importing the real reference (aidanmrli/rwm-pt-pytorch @ /root/reference) in
 * the build container -- tests/golden/generate_golden.py, fixtures under
 * tests/golden/ -- see tests/test_oracle_golden.py.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root).  Arithmetic is instantiated twice from
 * ptrwm_oracle_body.inc: REAL=float follows the reference's fp32 torch ops in
 * their source order; REAL=double is the same algorithm in fp64 ("truth").
 * Host pointers everywhere; the structs of include/ptrwm.h are reused so the
 * tests can drive oracle and HIP engine with one description.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ptrwm.h"

/* ---- Philox4x32-10 (Salmon et al., SC'11; Random123 philox.h) ------------
 * Not part of the reference (which precomputes torch.randn/torch.rand tensors,
 * rwm_gpu_optimized.py:490-511, pt_rwm_gpu_optimized.py:710-723); restated here
 * with the engine's counter layout so Philox-mode runs are reproducible on the
 * CPU.  Checked against the Random123 known-answer vectors. */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

/* word `w` of (stream, step, chain, temp): the engine's counter layout
 * (rwm-pt-pytorch_amd/csrc/philox.h header comment) */
static uint32_t philox_word(uint64_t seed, uint32_t stream, uint64_t step, uint64_t gchain, uint32_t t, uint32_t w) {
  uint32_t ctr[4], key[2], out[4];
  ctr[0] = (w >> 2) | ((uint32_t)(step >> 32) << 16);
  ctr[1] = (uint32_t)step;
  ctr[2] = (uint32_t)gchain;
  ctr[3] = t | (stream << 8) | ((uint32_t)(gchain >> 32) << 12);
  key[0] = (uint32_t)seed;
  key[1] = (uint32_t)(seed >> 32);
  oracle_philox4x32_10(ctr, key, out);
  return out[w & 3];
}

static float u01f(uint32_t r) { return (float)(r >> 8) * 0x1p-24f; }
static float u01_open0f(uint32_t r) { return ((float)(r >> 8) + 1.0f) * 0x1p-24f; }

int32_t oracle_ext_raw_per_step(int32_t kind, int32_t dim) {
  switch (kind) {
    case PTRWM_PROPOSAL_NORMAL: return dim;
    case PTRWM_PROPOSAL_LAPLACE: return dim;
    case PTRWM_PROPOSAL_UNIFORM_RADIUS: return dim + 1;
    default: return PTRWM_E_KIND;
  }
}

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define SFX _f32
#define R_EXP expf
#define R_LOG logf
#define R_LOG1P log1pf
#define R_SQRT sqrtf
#define R_POW powf
#define R_FABS fabsf
#include "ptrwm_oracle_body.inc"
#undef REAL
#undef SFX
#undef R_EXP
#undef R_LOG
#undef R_LOG1P
#undef R_SQRT
#undef R_POW
#undef R_FABS

#define REAL double
#define SFX _f64
#define R_EXP exp
#define R_LOG log
#define R_LOG1P log1p
#define R_SQRT sqrt
#define R_POW pow
#define R_FABS fabs
#include "ptrwm_oracle_body.inc"
#undef REAL
#undef SFX
#undef R_EXP
#undef R_LOG
#undef R_LOG1P
#undef R_SQRT
#undef R_POW
#undef R_FABS

/* The raw randoms oracle_run_f32 consumes in Philox mode (ext_prop == NULL), written out in the layout of the
 * external-randoms arrays of include/ptrwm.h, so that oracle_run_f32 driven with these arrays performs the SAME
 * float operations as its Philox mode (asserted bit for bit by tests/test_parity_checker.py).  This is what lets
 * tests/helpers.check_parity PROVE every decision on which the HIP kernel's in-kernel Philox path differs from the
 * oracle: the proof needs the step's proposal randoms and uniforms as numbers.
 *   ext_prop   [n_steps, n_chains, n_temps, raw]  NORMAL: the Box-Muller normals z; LAPLACE: the [0,1) uniforms;
 *                                                 UNIFORM_RADIUS: z[dim] then the radius uniform
 *   ext_u      [n_steps, n_chains, n_temps]       accept uniforms
 *   ext_swap_u [events, n_chains, n_temps - 1]    swap uniforms of the events of this step range (may be NULL)
 * Word map: rwm-pt-pytorch_amd/csrc/proposals.h header comment, as propose_one() above follows it. */
int32_t oracle_philox_randoms(int32_t proposal_kind, int32_t dim, int32_t n_temps, int64_t n_chains, int64_t chain_offset,
                              uint64_t seed, int64_t step0, int64_t n_steps, int64_t burn_in, int32_t swap_every,
                              float *ext_prop, float *ext_u, float *ext_swap_u) {
  const int n_raw = oracle_ext_raw_per_step(proposal_kind, dim);
  if (n_raw < 0) return n_raw;
  if (!ext_prop || !ext_u) return PTRWM_E_NULL;
  if (swap_every < 1) return PTRWM_E_ARG;
  const int T = n_temps, D = dim;
  const int w_even = 2 * ((D + 1) / 2);
  const int64_t se = swap_every;
  int64_t ev = 0; /* events of this call so far */
  for (int64_t i = 0; i < n_steps; ++i) {
    const uint64_t s = (uint64_t)(step0 + i);
    const int64_t sc = step0 + i + 1;
    const int swap_due = T > 1 && sc > burn_in && (sc % se == 0);
    for (int64_t c = 0; c < n_chains; ++c) {
      const uint64_t g = (uint64_t)(chain_offset + c);
      for (int t = 0; t < T; ++t) {
        const int64_t srep = (i * n_chains + c) * T + t;
        float *raw = ext_prop + srep * n_raw;
        if (proposal_kind == PTRWM_PROPOSAL_LAPLACE) {
          for (int d = 0; d < D; ++d) raw[d] = u01f(philox_word(seed, 0, s, g, (uint32_t)t, (uint32_t)d));
          ext_u[srep] = u01f(philox_word(seed, 0, s, g, (uint32_t)t, (uint32_t)D));
        } else {
          for (int d = 0; d < D; d += 2) {
            float z0, z1;
            box_muller_f32(philox_word(seed, 0, s, g, (uint32_t)t, (uint32_t)d),
                           philox_word(seed, 0, s, g, (uint32_t)t, (uint32_t)d + 1), &z0, &z1);
            raw[d] = z0;
            if (d + 1 < D) raw[d + 1] = z1;
          }
          if (proposal_kind == PTRWM_PROPOSAL_UNIFORM_RADIUS) {
            raw[D] = u01f(philox_word(seed, 0, s, g, (uint32_t)t, (uint32_t)w_even));
            ext_u[srep] = u01f(philox_word(seed, 0, s, g, (uint32_t)t, (uint32_t)w_even + 1));
          } else {
            ext_u[srep] = u01f(philox_word(seed, 0, s, g, (uint32_t)t, (uint32_t)w_even));
          }
        }
      }
      if (swap_due && ext_swap_u)
        for (int j = 0; j < T - 1; ++j)
          ext_swap_u[(ev * n_chains + c) * (T - 1) + j] = u01f(philox_word(seed, 1, s, g, (uint32_t)j, 0));
    }
    if (swap_due) ++ev;
  }
  return PTRWM_OK;
}
