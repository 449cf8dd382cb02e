// hash_ablate.hip -- why does k_post (SHA3-256 of a 1088-byte ciphertext + KDF, ten
// Keccak-f per instance) run at 8.2-8.5 G permutations/s when the bare permutation
// sustains 10.1 G/s?  The product kernel next to two variants of itself:
//   same-ct   every work-item hashes instance 0's ciphertext (loads hit L1/L2)
//   no-loads  the absorbed words are synthesised in registers
#include "../../kyber_amd/csrc/kem_kernels.h"
#include <cstdio>
#include "../../include/kyber_batch.h"   // the last section runs the product pipeline (link with -lkyber_hip)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace ky;
using P = Par<3>;

template <int MODE>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_post_var(int n, const uint8_t* __restrict__ ct, const uint8_t* __restrict__ small, uint8_t* __restrict__ ss) {
  int i = blockIdx.x * SPONGE_WG + threadIdx.x;
  if (i >= n) return;
  u64p A[25], kb[4], h[4];
  if constexpr (MODE < 2) {
    keccak_absorb_words<136, 0x06, P::CTB>(A, ct + (MODE == 1 ? 0 : (size_t)i * P::CTB));
  } else {
    keccak_zero(A);
#pragma unroll 1
    for (int b = 0; b < 9; b++) {
#pragma unroll
      for (int w = 0; w < 17; w++) A[w] ^= u64p{(uint32_t)(i + w), (uint32_t)(b * 17 + w)};
      keccak_f1600(A);
    }
  }
#pragma unroll
  for (int w = 0; w < 4; w++) h[w] = A[w];
  ld32B(small + (size_t)i * SMALL + 32, kb);
  shake256_64(A, kb, h);
  st32B(ss + (size_t)i * 32, A[0], A[1], A[2], A[3]);
}

// something to put the chip into the state the pipeline leaves it in: fp32 FMAs on every lane for ~1.5 ms
__global__ void __launch_bounds__(256) k_burn(float* out, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f, d = a + 1, e = a + 2, f = a + 3;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) { a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c); e = __builtin_fmaf(e, b, c); f = __builtin_fmaf(f, b, c); }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + d + e + f;
}
int g_reps = 4;
template <int MODE> float run_after_burn(int n, const uint8_t* ct, const uint8_t* small, uint8_t* ss, float* scratch) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < g_reps; r++) {
    if (r == g_reps - 50) best = 1e30f;   // sustained runs: report the best of the last 50
    k_burn<<<256 * 16, 256>>>(scratch, 3000);
    hipEventRecord(a);
    k_post_var<MODE><<<(n + SPONGE_WG - 1) / SPONGE_WG, SPONGE_WG>>>(n, ct, small, ss);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}
template <int MODE> float run(int n, const uint8_t* ct, const uint8_t* small, uint8_t* ss) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < 4; r++) {
    hipEventRecord(a);
    k_post_var<MODE><<<(n + SPONGE_WG - 1) / SPONGE_WG, SPONGE_WG>>>(n, ct, small, ss);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int n = 1 << 19;
  uint8_t *ct, *small, *ss;
  CK(hipMalloc(&ct, (size_t)n * P::CTB)); CK(hipMalloc(&small, (size_t)n * 128)); CK(hipMalloc(&ss, (size_t)n * 32));
  CK(hipMemset(ct, 0x5a, (size_t)n * P::CTB)); CK(hipMemset(small, 0x33, (size_t)n * 128));
  for (int r = 0; r < 3; r++) {
    float t0 = run<0>(n, ct, small, ss), t1 = run<1>(n, ct, small, ss), t2 = run<2>(n, ct, small, ss);
    printf("n = %d: product %.3f ms = %.2f G perm/s | same-ct %.3f ms = %.2f | no-loads %.3f ms = %.2f\n", n, t0, n * 10.0 / t0 * 1e-6, t1,
           n * 10.0 / t1 * 1e-6, t2, n * 10.0 / t2 * 1e-6);
  }
  float* scratch; CK(hipMalloc(&scratch, 256 * 16 * 256 * 4));
  for (int r = 0; r < 3; r++) {
    float t0 = run_after_burn<0>(n, ct, small, ss, scratch), t2 = run_after_burn<2>(n, ct, small, ss, scratch);
    printf("right after 1.5 ms of fp32 FMAs on every lane: product %.3f ms = %.2f G perm/s | no-loads %.3f ms = %.2f\n", t0, n * 10.0 / t0 * 1e-6, t2,
           n * 10.0 / t2 * 1e-6);
  }
  g_reps = 400;   // ~0.8 s of back-to-back work per measurement
  for (int r = 0; r < 2; r++) {
    float t0 = run_after_burn<0>(n, ct, small, ss, scratch), t2 = run_after_burn<2>(n, ct, small, ss, scratch);
    printf("the same, sustained for 400 rounds (best of the last 50): product %.3f ms = %.2f G perm/s | no-loads %.3f ms = %.2f\n", t0, n * 10.0 / t0 * 1e-6, t2,
           n * 10.0 / t2 * 1e-6);
  }
  // the same kernel timed inside the product pipeline's sequence: keypair once, then encaps + decaps
  // of 2^19 instances back to back, the variant launched right behind them on the same stream
  {
    uint8_t *pk, *sk, *kc, *ec, *ss2;
    CK(hipMalloc(&pk, (size_t)n * 1184)); CK(hipMalloc(&sk, (size_t)n * 2400)); CK(hipMalloc(&kc, (size_t)n * 64)); CK(hipMalloc(&ec, (size_t)n * 32));
    CK(hipMalloc(&ss2, (size_t)n * 32));
    CK(hipMemset(kc, 0x11, (size_t)n * 64)); CK(hipMemset(ec, 0x22, (size_t)n * 32));
    if (kyber_batch_keypair_dev(3, 0, n, pk, sk, kc, nullptr)) { printf("keypair failed\n"); return 1; }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int r = 0; r < 6; r++) {
      if (kyber_batch_enc_dev(3, 0, n, ct, ss, pk, ec, nullptr) || kyber_batch_dec_dev(3, 0, n, ss2, ct, sk, nullptr)) { printf("enc/dec failed\n"); return 1; }
      CK(hipEventRecord(a));
      k_post_var<0><<<(n + SPONGE_WG - 1) / SPONGE_WG, SPONGE_WG>>>(n, ct, small, ss);
      CK(hipEventRecord(b));
      k_post_var<2><<<(n + SPONGE_WG - 1) / SPONGE_WG, SPONGE_WG>>>(n, ct, small, ss);
      hipEvent_t c; CK(hipEventCreate(&c)); CK(hipEventRecord(c)); CK(hipEventSynchronize(c));
      float t0, t2; CK(hipEventElapsedTime(&t0, a, b)); CK(hipEventElapsedTime(&t2, b, c));
      printf("right behind encaps + decaps of the product: product %.3f ms = %.2f G perm/s | no-loads %.3f ms = %.2f\n", t0, n * 10.0 / t0 * 1e-6, t2,
             n * 10.0 / t2 * 1e-6);
    }
  }
  return 0;
}
