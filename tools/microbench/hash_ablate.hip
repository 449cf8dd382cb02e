// hash_ablate.hip -- why does k_post (SHA3-256 of a 1088-byte ciphertext + KDF, ten
// Keccak-f per instance) run at 8.2-8.5 G permutations/s when the bare permutation
// sustains 10.1 G/s?  The product kernel next to two variants of itself:
//   same-ct   every work-item hashes instance 0's ciphertext (loads hit L1/L2)
//   no-loads  the absorbed words are synthesised in registers
//   16-byte loads (timing only) and workgroup-coalesced loads through LDS: the last section
#include "../../kyber_amd/csrc/kem_kernels.h"
#include <cstdio>
#include <type_traits>
#include "../../include/kyber_batch.h"   // the last section runs the product pipeline (link with -lkyber_hip)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace ky;
using P = Par<3>;

// Tried and measured here, not in the product (0.69 ms against 0.58):
// the absorb loop for a whole workgroup at once, with coalesced loads: lane t hashes the LEN bytes at
// base + t * stride (t < nvalid; the other lanes run along on lane nvalid-1's message).  A per-lane
// read of a 136-byte block touches two or three 128-byte lines per lane, 64 lines per load
// instruction; here the workgroup fetches the W words of all its lanes' blocks as one flat array
// (entry f = r * W + w, lane l takes f = l, l + WG, ...: consecutive lanes read consecutive words),
// parks them in LDS and every lane picks up its own W words.  EVERY wavefront of the workgroup must
// call it (no early exit in front of it); lds holds WG * RATE / 8 words.
template <int RATE, int DOM, int LEN>
__device__ __forceinline__ void keccak_absorb_coop(u64p (&A)[25], const uint8_t* base, size_t stride, int nvalid, uint2* lds) {
  static_assert(LEN % 8 == 0 && RATE % 8 == 0, "word-aligned use only");
  constexpr int RW = RATE / 8, NFULL = LEN / RATE, TAILW = (LEN % RATE) / 8;
  const int tid = threadIdx.x, wg = blockDim.x;
  uint2 g[RW];
  auto fetch = [&](int word0, auto nwords) {   // request words word0 .. word0+W-1 of every lane's message
    constexpr int W = decltype(nwords)::value;
#pragma unroll
    for (int j = 0; j < W; j++) {
      const int f = tid + wg * j, r0 = f / W, w = f - r0 * W, r = r0 < nvalid ? r0 : nvalid - 1;
      g[j] = *reinterpret_cast<const uint2*>(base + (size_t)r * stride + 8 * (word0 + w));
    }
  };
  auto exchange = [&](auto nwords) {           // g (flat order) -> LDS -> g (this lane's own words)
    constexpr int W = decltype(nwords)::value;
#pragma unroll
    for (int j = 0; j < W; j++) lds[tid + wg * j] = g[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < W; j++) g[j] = lds[tid * W + j];
    __syncthreads();
  };
  keccak_zero(A);
  if constexpr (NFULL > 0) {
    fetch(0, std::integral_constant<int, RW>{});
#pragma unroll 1
    for (int b = 0; b < NFULL; b++) {
      exchange(std::integral_constant<int, RW>{});
#pragma unroll
      for (int w = 0; w < RW; w++) A[w] ^= u64p{g[w].x, g[w].y};
      if (b + 1 < NFULL) fetch(RW * (b + 1), std::integral_constant<int, RW>{});   // uniform branch: in flight during the permutation
      else if constexpr (TAILW > 0) fetch(RW * NFULL, std::integral_constant<int, TAILW>{});
      keccak_f1600(A);
    }
  } else if constexpr (TAILW > 0) {
    fetch(0, std::integral_constant<int, TAILW>{});
  }
  if constexpr (TAILW > 0) {
    exchange(std::integral_constant<int, TAILW>{});
#pragma unroll
    for (int w = 0; w < TAILW; w++) A[w] ^= u64p{g[w].x, g[w].y};
  }
  A[TAILW].lo ^= (uint32_t)DOM;
  A[RW - 1].hi ^= 0x80000000u;
  keccak_f1600(A);
}



template <int MODE>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_post_var(int n, const uint8_t* __restrict__ ct, const uint8_t* __restrict__ small, uint8_t* __restrict__ ss) {
  int i = blockIdx.x * SPONGE_WG + threadIdx.x;
  if (MODE != 4 && i >= n) return;
  u64p A[25], kb[4], h[4];
  if constexpr (MODE < 2) {
    keccak_absorb_words<136, 0x06, P::CTB>(A, ct + (MODE == 1 ? 0 : (size_t)i * P::CTB));
  } else if constexpr (MODE == 4) {
    __shared__ uint2 xch[SPONGE_WG * 17];
    const int first = blockIdx.x * SPONGE_WG, nvalid = n - first < SPONGE_WG ? n - first : SPONGE_WG;
    keccak_absorb_coop<136, 0x06, P::CTB>(A, ct + (size_t)first * P::CTB, P::CTB, nvalid, xch);
  } else if constexpr (MODE == 3) {
    // timing only: nine 16-byte loads per block at 16-byte-aligned offsets (block b starts at 136 b,
    // rounded down), next block requested before the permutation, the first 17 words absorbed
    const uint8_t* in = ct + (size_t)i * P::CTB;
    keccak_zero(A);
    uint4 nx[9];
#pragma unroll
    for (int q = 0; q < 9; q++) nx[q] = reinterpret_cast<const uint4*>(in)[q];
#pragma unroll 1
    for (int b = 0; b < 8; b++) {
#pragma unroll
      for (int q = 0; q < 8; q++) { A[2 * q] ^= u64p{nx[q].x, nx[q].y}; A[2 * q + 1] ^= u64p{nx[q].z, nx[q].w}; }
      A[16] ^= u64p{nx[8].x, nx[8].y};
      const int nb = b + 1 < 8 ? b + 1 : b;
      const uint4* src = reinterpret_cast<const uint4*>(in + ((136 * nb) & ~15));
      const int lim = (P::CTB - ((136 * nb) & ~15)) / 16;   // stay inside this instance's bytes
#pragma unroll
      for (int q = 0; q < 9; q++) nx[q] = src[q < lim ? q : lim - 1];
      keccak_f1600(A);
    }
    A[0].lo ^= 0x06; A[16].hi ^= 0x80000000u;
    keccak_f1600(A);
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
  if (i >= n) i = n - 1;
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
      hipEvent_t c, d; CK(hipEventCreate(&c)); CK(hipEventCreate(&d)); CK(hipEventRecord(c));
      k_post_var<3><<<(n + SPONGE_WG - 1) / SPONGE_WG, SPONGE_WG>>>(n, ct, small, ss);
      CK(hipEventRecord(d));
      k_post_var<4><<<(n + SPONGE_WG - 1) / SPONGE_WG, SPONGE_WG>>>(n, ct, small, ss);
      hipEvent_t ev5; CK(hipEventCreate(&ev5)); CK(hipEventRecord(ev5)); CK(hipEventSynchronize(ev5));
      float t0, t2, t3, t4; CK(hipEventElapsedTime(&t0, a, b)); CK(hipEventElapsedTime(&t2, b, c)); CK(hipEventElapsedTime(&t3, c, d)); CK(hipEventElapsedTime(&t4, d, ev5));
      printf("right behind encaps + decaps of the product: product %.3f ms = %.2f G perm/s | no-loads %.3f ms = %.2f | 16-byte loads %.3f ms = %.2f | workgroup-coalesced loads %.3f ms = %.2f\n", t0,
             n * 10.0 / t0 * 1e-6, t2, n * 10.0 / t2 * 1e-6, t3, n * 10.0 / t3 * 1e-6, t4, n * 10.0 / t4 * 1e-6);
    }
  }
  return 0;
}
