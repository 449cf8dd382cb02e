// keccak.h -- Keccak-f[1600] and sponge helpers, one sponge per work-item.
//
// Mapping (gfx950): the 25 x 64-bit lanes of one sponge live in 50 VGPRs of
// one work-item as explicit (lo,hi) 32-bit halves; a 64-wide wavefront
// advances 64 independent sponges with no cross-lane traffic.  Every array
// index is a compile-time constant after unrolling, so the state never leaves
// registers.  Instruction selection is pinned by hand:
//   * theta parity and the theta-apply xor are 3-input xors  -> v_bitop3_b32 (0x96)
//   * chi  a ^ (~b & c) is one 3-input boolean               -> v_bitop3_b32 (0xD2)
//   * 64-bit rotates are two v_alignbit_b32 (halves swapped when n > 32)
// which gives 180 VALU ops per round (20 + 10 + 50 + 48 + 50 + 2) = 4320 per
// permutation; nothing here is 64-bit arithmetic, so nothing is rate-limited.
//
// Replaces (behaviour, not text): ref/fips202.c:82-344 KeccakF1600_StatePermute,
// :461-485 keccak_absorb_once, :500-514 keccak_squeezeblocks, :745-774 sha3_*.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "consts_gen.h"

namespace ky {

__constant__ const uint64_t KECCAK_RC[24] = KY_KECCAK_RC_INIT;

struct u64p {
  uint32_t lo, hi;
};

__device__ __forceinline__ u64p mk64(uint64_t v) { return u64p{(uint32_t)v, (uint32_t)(v >> 32)}; }
__device__ __forceinline__ uint64_t to64(u64p v) { return (uint64_t)v.hi << 32 | v.lo; }
__device__ __forceinline__ u64p operator^(u64p a, u64p b) { return u64p{a.lo ^ b.lo, a.hi ^ b.hi}; }
__device__ __forceinline__ u64p& operator^=(u64p& a, u64p b) {
  a.lo ^= b.lo;
  a.hi ^= b.hi;
  return a;
}

template <int TT>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
  return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, TT);
}
__device__ __forceinline__ u64p xor3(u64p a, u64p b, u64p c) {
  return u64p{bitop3<0x96>(a.lo, b.lo, c.lo), bitop3<0x96>(a.hi, b.hi, c.hi)};
}
// b0 ^ (~b1 & b2)
__device__ __forceinline__ u64p chi(u64p b0, u64p b1, u64p b2) {
  return u64p{bitop3<0xD2>(b0.lo, b1.lo, b2.lo), bitop3<0xD2>(b0.hi, b1.hi, b2.hi)};
}

template <int N>
__device__ __forceinline__ u64p rol(u64p v) {
  if constexpr (N == 0) return v;
  else if constexpr (N == 32) return u64p{v.hi, v.lo};
  else if constexpr (N < 32)
    return u64p{__builtin_amdgcn_alignbit(v.lo, v.hi, 32 - N), __builtin_amdgcn_alignbit(v.hi, v.lo, 32 - N)};
  else
    return u64p{__builtin_amdgcn_alignbit(v.hi, v.lo, 64 - N), __builtin_amdgcn_alignbit(v.lo, v.hi, 64 - N)};
}

// rho offset of lane (x,y): FIPS 202 §3.2.2, t-th step of the (x,y)->(y,2x+3y) walk
__host__ __device__ constexpr int keccak_rho(int x, int y) {
  int cx = 1, cy = 0;
  for (int t = 0; t < 24; t++) {
    if (cx == x && cy == y) return ((t + 1) * (t + 2) / 2) % 64;
    int ny = (2 * cx + 3 * cy) % 5;
    cx = cy;
    cy = ny;
  }
  return 0;
}

// One round, A -> E (separate arrays so pi is pure renaming).
//
// Phase lock.  gfx950 issues the logic ops (v_bitop3_b32, 2 cycles per wavefront) and the rotates
// (v_alignbit_b32, 4 cycles) at those rates only while the wavefronts that share a SIMD issue the same
// class: mixed streams cost ~0.9 cycle more per rotate whatever the run lengths inside one wavefront
// (tools/microbench/valu_rate.hip "mix run N"), because the wavefronts drift apart and the SIMD
// interleaves them.  So a round is cut into five single-class phases (20 + 10 + 50 + 48 + 50
// instructions) and the wavefronts of a workgroup enter each phase together: kernels that hash run
// 512-work-item workgroups (SPONGE_WG), two wavefronts per SIMD from one workgroup, two workgroups per
// CU.  Bare permutation 10.1 -> 11.9 G/s (tools/microbench/keccak_rate.hip, profiles/
// r02_microbench_keccak_rate.txt; barriers per round or per two rounds instead lose to no barrier).
// The barrier only aligns issue: no data crosses it, waves that have ended do not count, and a
// wavefront that runs more permutations than its neighbours pairs its barriers with whoever is left.
// It is written as asm so that the compiler does not drain the memory counters in front of it.
// For the same reason a kernel that calls the permutation must not ALSO use __syncthreads() for data
// (none does: sponge kernels share nothing between lanes): a wavefront inside a permutation would
// answer the data barrier of one that is not.
constexpr int SPONGE_WG = 512;
// (hipcc's second __launch_bounds__ argument is wavefronts per SIMD, not workgroups per CU: the sponge
// kernels ask for 4 = two 512-lane workgroups per CU, i.e. at most 128 VGPRs)
// Below SPONGE_WG_MIN_ITEMS sponges a launch cannot fill the chip with 512-lane workgroups (one
// per CU = 131072 lanes): it then runs one wavefront per workgroup, which spreads the wavefronts
// over all SIMDs (the barrier of a one-wavefront workgroup is a no-op).  Kernels index with
// blockDim.x, so the host picks the size per launch.
constexpr int SPONGE_WG_SMALL = 64;
constexpr size_t SPONGE_WG_MIN_ITEMS = 131072;
__host__ __device__ constexpr unsigned sponge_wg(size_t items) { return items >= SPONGE_WG_MIN_ITEMS ? SPONGE_WG : SPONGE_WG_SMALL; }
// Lock modes: 0 = the compiler's own schedule, 2 = the phase lock (product).  tools/microbench also
// builds 1 (phases, no barrier), 3 / 4 (a barrier per round / per two rounds) and 5 / 6 (four / two
// barriers per round) through -DKY_KECCAK_PHASED=<mode>.  Small launches (one-wavefront
// workgroups) keep mode 2: the matrix sampler instantiated with mode 0 for them changed nothing
// measurable at 256 and 4096 instances per call.
#if defined(KY_KECCAK_PHASED)
constexpr int KY_LOCK_MODE = KY_KECCAK_PHASED;
#else
constexpr int KY_LOCK_MODE = 2;
#endif
__device__ __forceinline__ void ky_sync() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier");
  __builtin_amdgcn_sched_barrier(0);
}
// boundary n of the round: 1 after the parities, 2 after their rotates, 3 after theta, 4 after rho, 5 after chi
template <int MODE, int N>
__device__ __forceinline__ void ky_phase() {
  if constexpr (MODE == 0) return;
  else if constexpr (MODE == 2 || (MODE == 5 && N != 5) || (MODE == 6 && (N == 3 || N == 4))) ky_sync();
  else __builtin_amdgcn_sched_barrier(0);
}
template <int MODE>
__device__ __forceinline__ void keccak_round(const u64p (&A)[25], u64p (&E)[25], uint64_t rc) {
  u64p C[5], R[5], T[25], B[25];
  if constexpr (MODE == 3) ky_sync();
#pragma unroll
  for (int x = 0; x < 5; x++) C[x] = xor3(xor3(A[x], A[x + 5], A[x + 10]), A[x + 15], A[x + 20]);
  ky_phase<MODE, 1>();
#pragma unroll
  for (int x = 0; x < 5; x++) R[x] = rol<1>(C[x]);
  ky_phase<MODE, 2>();
  // theta-apply, then rho/pi: B[y, 2x+3y] = rol(A[x,y] ^ C[x-1] ^ rol1(C[x+1]), rho(x,y))
#pragma unroll
  for (int y = 0; y < 5; y++)
#pragma unroll
    for (int x = 0; x < 5; x++) T[x + 5 * y] = xor3(A[x + 5 * y], C[(x + 4) % 5], R[(x + 1) % 5]);
  ky_phase<MODE, 3>();
#define KY_RPI(x, y) B[y + 5 * ((2 * x + 3 * y) % 5)] = rol<keccak_rho(x, y)>(T[x + 5 * y]);
#define KY_RPI_ROW(y) KY_RPI(0, y) KY_RPI(1, y) KY_RPI(2, y) KY_RPI(3, y) KY_RPI(4, y)
  KY_RPI_ROW(0) KY_RPI_ROW(1) KY_RPI_ROW(2) KY_RPI_ROW(3) KY_RPI_ROW(4)
#undef KY_RPI_ROW
#undef KY_RPI
  ky_phase<MODE, 4>();
#pragma unroll
  for (int y = 0; y < 5; y++)
#pragma unroll
    for (int x = 0; x < 5; x++)
      E[x + 5 * y] = chi(B[x + 5 * y], B[(x + 1) % 5 + 5 * y], B[(x + 2) % 5 + 5 * y]);
  E[0].lo ^= (uint32_t)rc;
  E[0].hi ^= (uint32_t)(rc >> 32);
  ky_phase<MODE, 5>();
}

// LOCK = false: the compiler's schedule (microbenchmarks)
template <bool LOCK = true>
__device__ __forceinline__ void keccak_f1600(u64p (&A)[25]) {
  constexpr int MODE = LOCK ? KY_LOCK_MODE : 0;
  u64p E[25];
#pragma unroll 1
  for (int r = 0; r < 24; r += 2) {
    if constexpr (MODE == 4) ky_sync();
    keccak_round<MODE>(A, E, KECCAK_RC[r]);
    keccak_round<MODE>(E, A, KECCAK_RC[r + 1]);
  }
}

__device__ __forceinline__ void keccak_zero(u64p (&A)[25]) {
#pragma unroll
  for (int i = 0; i < 25; i++) A[i] = u64p{0u, 0u};
}

// 8-byte little-endian loads/stores on byte streams with 8-byte aligned bases
// (every KEM object stride is a multiple of 16, see kyber_batch.h).
__device__ __forceinline__ u64p ld64(const uint8_t* p) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  return u64p{v.x, v.y};
}
__device__ __forceinline__ void st64(uint8_t* p, u64p v) { *reinterpret_cast<uint2*>(p) = make_uint2(v.lo, v.hi); }

// Absorb LEN bytes (multiple of 8, 8-byte aligned) starting from a zero state,
// pad with domain byte DOM at rate RATE and run the final permutation, leaving
// the first output block in A.  Restates keccak_absorb_once + the first
// squeeze permutation (ref/fips202.c:461-485, :508).
template <int RATE, int DOM, int LEN>
__device__ __forceinline__ void keccak_absorb_words(u64p (&A)[25], const uint8_t* in) {
  static_assert(LEN % 8 == 0 && RATE % 8 == 0, "word-aligned use only");
  constexpr int RW = RATE / 8, NFULL = LEN / RATE, TAILW = (LEN % RATE) / 8;
  keccak_zero(A);
  if constexpr (NFULL > 0) {
    // The words of block b+1 are requested before block b is permuted.  Every wavefront
    // of these kernels does the same work at the same pace, so without this they all sit
    // in their load phase together and the SIMD idles for a memory round trip per block.
    // The last pass re-reads the last full block (index clamped: no load under a branch,
    // so the wait in front of the xor stays counted); TAILW words are fetched afterwards.
    u64p nx[RW];
#pragma unroll
    for (int w = 0; w < RW; w++) nx[w] = ld64(in + 8 * w);
#pragma unroll 1
    for (int b = 0; b < NFULL; b++) {
#pragma unroll
      for (int w = 0; w < RW; w++) A[w] ^= nx[w];
      const uint8_t* nb = in + 8 * RW * (b + 1 < NFULL ? b + 1 : b);
#pragma unroll
      for (int w = 0; w < RW; w++) nx[w] = ld64(nb + 8 * w);
      keccak_f1600(A);
    }
  }
#pragma unroll
  for (int w = 0; w < TAILW; w++) A[w] ^= ld64(in + 8 * (NFULL * RW + w));
  A[TAILW].lo ^= (uint32_t)DOM;
  A[RW - 1].hi ^= 0x80000000u;
  keccak_f1600(A);
}

}  // namespace ky
