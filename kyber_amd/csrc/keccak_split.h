// keccak_split.h -- the FO hashes for SMALL batches: one sponge on TWO adjacent work-items.
//
// Up to a few thousand instances no kernel fills the chip, and a call's time is the chain of
// dependent permutations of its longest lane: a wavefront alone on its SIMD issues one instruction
// every ~4.7 cycles whatever the instruction, so the 4320 instructions of keccak.h's permutation take
// ~9.5 us -- 25 of them end to end in an encapsulation ("small batches").  Here the even
// work-item of a pair holds the LOW 32-bit halves of the 25 lanes and the odd one the HIGH halves, in
// the standard representation (no bit interleaving, so messages and digests are read and written as
// they lie in memory, each work-item its own dword of every 8 bytes):
//   * theta's parities, theta-apply and chi are bitwise: each half on its own, half the instructions;
//   * a 64-bit rotate by n is own' = alignbit(own, other, 32 - n) for n < 32 and
//     alignbit(other, own, 64 - n) for n > 32 -- the SAME instruction on both work-items once the
//     partner's half has been fetched through the DPP crossbar (quad_perm [1,0,3,2], one v_mov_b32_dpp).
// 10 + 5 + 5 + 25 + 24 + 24 + 25 + 2 = 120 instructions per round and wavefront instead of 180: a
// permutation in two thirds of the time, for twice the lanes -- which small batches have to spare.
// Large batches keep keccak.h's layout (one sponge per work-item, no cross-lane traffic, phase lock).
//
// Same functions restated as keccak.h / kem_kernels.h: ref/fips202.c:82-344, :461-485, :745-774;
// ref/kem.c:23-127 (the hashes around the IND-CPA core).
#pragma once
#include "kem_kernels.h"

namespace ky {

// lanes per sponge-pair workgroup: one wavefront = 32 sponges
constexpr int SPLIT_WG = 64;
// Batches up to this many instances hash on pairs.  Same-box A/B of the threshold, device-pointer calls
// (profiles/r03_ab_midsize_thresholds.txt): n = 32768 enc 460 -> 437 us; 65536 a tie; 131072 +9 %: from one
// wavefront per SIMD on, the pair's 240 lane-instructions per round cost more than its shorter chain saves.
#ifndef KY_SPLIT_MAX_N
#define KY_SPLIT_MAX_N 32768
#endif
constexpr size_t SPLIT_MAX_N = KY_SPLIT_MAX_N;

__device__ __forceinline__ uint32_t ks_other(uint32_t x) {  // the partner's copy of x
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);
}
// this work-item's half of rol64(v, N), v = (own, other) or (other, own)
template <int N>
__device__ __forceinline__ uint32_t ks_rol(uint32_t own, uint32_t oth) {
  if constexpr (N == 0) return own;
  else if constexpr (N == 32) return oth;
  else if constexpr (N < 32) return __builtin_amdgcn_alignbit(own, oth, 32 - N);
  else return __builtin_amdgcn_alignbit(oth, own, 64 - N);
}

__device__ __forceinline__ void ks_round(const uint32_t (&A)[25], uint32_t (&E)[25], uint32_t rc) {
  uint32_t C[5], R[5], T[25], B[25];
#pragma unroll
  for (int x = 0; x < 5; x++) C[x] = bitop3<0x96>(bitop3<0x96>(A[x], A[x + 5], A[x + 10]), A[x + 15], A[x + 20]);
#pragma unroll
  for (int x = 0; x < 5; x++) R[x] = ks_rol<1>(C[x], ks_other(C[x]));
#pragma unroll
  for (int y = 0; y < 5; y++)
#pragma unroll
    for (int x = 0; x < 5; x++) T[x + 5 * y] = bitop3<0x96>(A[x + 5 * y], C[(x + 4) % 5], R[(x + 1) % 5]);
#define KS_RPI(x, y) B[y + 5 * ((2 * x + 3 * y) % 5)] = ks_rol<keccak_rho(x, y)>(T[x + 5 * y], keccak_rho(x, y) ? ks_other(T[x + 5 * y]) : 0u);
#define KS_RPI_ROW(y) KS_RPI(0, y) KS_RPI(1, y) KS_RPI(2, y) KS_RPI(3, y) KS_RPI(4, y)
  KS_RPI_ROW(0) KS_RPI_ROW(1) KS_RPI_ROW(2) KS_RPI_ROW(3) KS_RPI_ROW(4)
#undef KS_RPI_ROW
#undef KS_RPI
#pragma unroll
  for (int y = 0; y < 5; y++)
#pragma unroll
    for (int x = 0; x < 5; x++)
      E[x + 5 * y] = bitop3<0xD2>(B[x + 5 * y], B[(x + 1) % 5 + 5 * y], B[(x + 2) % 5 + 5 * y]);
  E[0] ^= rc;
}

// hi = all ones on the work-item that holds the high halves, 0 on the other
struct KsHalf {
  uint32_t hi, off;  // off = 4 on the high work-item: byte offset of its dword in every 8 bytes
  __device__ __forceinline__ uint32_t low(uint32_t v) const { return v & ~hi; }   // v in the low half, 0 in the high
  __device__ __forceinline__ uint32_t high(uint32_t v) const { return v & hi; }
};
__device__ __forceinline__ KsHalf ks_half(int p) { return KsHalf{0u - (uint32_t)p, 4u * (uint32_t)p}; }

__device__ __forceinline__ void ks_f1600(uint32_t (&A)[25], KsHalf h) {
  uint32_t E[25];
#pragma unroll 1
  for (int r = 0; r < 24; r += 2) {
    const uint64_t c0 = KECCAK_RC[r], c1 = KECCAK_RC[r + 1];
    ks_round(A, E, (uint32_t)c0 ^ (h.hi & ((uint32_t)c0 ^ (uint32_t)(c0 >> 32))));
    ks_round(E, A, (uint32_t)c1 ^ (h.hi & ((uint32_t)c1 ^ (uint32_t)(c1 >> 32))));
  }
}
__device__ __forceinline__ void ks_zero(uint32_t (&A)[25]) {
#pragma unroll
  for (int i = 0; i < 25; i++) A[i] = 0u;
}
__device__ __forceinline__ uint32_t ks_ld(const uint8_t* p, KsHalf h) { return *reinterpret_cast<const uint32_t*>(p + h.off); }
// this work-item's halves of four 8-byte words
__device__ __forceinline__ void ks_ld32B(const uint8_t* p, KsHalf h, uint32_t (&v)[4]) {
#pragma unroll
  for (int w = 0; w < 4; w++) v[w] = ks_ld(p + 8 * w, h);
}
__device__ __forceinline__ void ks_st32B(uint8_t* p, KsHalf h, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  uint32_t* q = reinterpret_cast<uint32_t*>(p + h.off);
  q[0] = a; q[2] = b; q[4] = c; q[6] = d;
}

// keccak_absorb_words of keccak.h on a pair
template <int RATE, int DOM, int LEN>
__device__ __forceinline__ void ks_absorb_words(uint32_t (&A)[25], const uint8_t* in, KsHalf h) {
  static_assert(LEN % 8 == 0 && RATE % 8 == 0, "word-aligned use only");
  constexpr int RW = RATE / 8, NFULL = LEN / RATE, TAILW = (LEN % RATE) / 8;
  ks_zero(A);
  if constexpr (NFULL > 0) {
    uint32_t nx[RW];
#pragma unroll
    for (int w = 0; w < RW; w++) nx[w] = ks_ld(in + 8 * w, h);
#pragma unroll 1
    for (int b = 0; b < NFULL; b++) {
#pragma unroll
      for (int w = 0; w < RW; w++) A[w] ^= nx[w];
      const uint8_t* nb = in + 8 * RW * (b + 1 < NFULL ? b + 1 : b);
#pragma unroll
      for (int w = 0; w < RW; w++) nx[w] = ks_ld(nb + 8 * w, h);
      ks_f1600(A, h);
    }
  }
#pragma unroll
  for (int w = 0; w < TAILW; w++) A[w] ^= ks_ld(in + 8 * (NFULL * RW + w), h);
  A[TAILW] ^= h.low((uint32_t)DOM);
  A[RW - 1] ^= h.high(0x80000000u);
  ks_f1600(A, h);
}
// sha3_256 / sha3_512 / shake256 of 32 or 64 bytes held in registers (kem_kernels.h:27-52)
__device__ __forceinline__ void ks_sha3_256_32(uint32_t (&A)[25], const uint32_t (&in)[4], KsHalf h) {
  ks_zero(A);
#pragma unroll
  for (int w = 0; w < 4; w++) A[w] = in[w];
  A[4] = h.low(0x06);
  A[16] = h.high(0x80000000u);
  ks_f1600(A, h);
}
template <uint32_t DOM, int LASTW>
__device__ __forceinline__ void ks_hash64(uint32_t (&A)[25], const uint32_t (&a)[4], const uint32_t (&b)[4], KsHalf h) {
  ks_zero(A);
#pragma unroll
  for (int w = 0; w < 4; w++) { A[w] = a[w]; A[4 + w] = b[w]; }
  A[8] = h.low(DOM);
  A[LASTW] ^= h.high(0x80000000u);
  ks_f1600(A, h);
}
__device__ __forceinline__ void ks_sha3_512_64(uint32_t (&A)[25], const uint32_t (&a)[4], const uint32_t (&b)[4], KsHalf h) {
  ks_hash64<0x06, 8>(A, a, b, h);  // rate 72
}
__device__ __forceinline__ void ks_shake256_64(uint32_t (&A)[25], const uint32_t (&a)[4], const uint32_t (&b)[4], KsHalf h) {
  ks_hash64<0x1F, 16>(A, a, b, h);  // rate 136
}

// ------------------------------------------------------------------ the FO hashes on pairs
// Work-item g: instance g >> 1, half g & 1.  Same arguments, same bytes as the kernels of kem_kernels.h
// they stand in for (k_kp_pre, k_kp_post, k_hash_pk, k_enc_pre, k_dec_mid, k_post).
#define KS_PROLOGUE(n_)                                              \
  const int g_ = blockIdx.x * SPLIT_WG + threadIdx.x, i = g_ >> 1;   \
  if (i >= (n_)) return;                                             \
  const KsHalf h = ks_half(g_ & 1)

__global__ void __launch_bounds__(SPLIT_WG) k2_kp_pre(int n, const uint8_t* __restrict__ coins, uint8_t* __restrict__ small) {
  KS_PROLOGUE(n);
  uint32_t d[4], A[25];
  ks_ld32B(coins + (size_t)i * 64, h, d);
  ks_zero(A);
#pragma unroll
  for (int w = 0; w < 4; w++) A[w] = d[w];
  A[4] = h.low(0x06);
  A[8] = h.high(0x80000000u);
  ks_f1600(A, h);
  uint8_t* s = small + (size_t)i * SMALL;
  ks_st32B(s, h, A[0], A[1], A[2], A[3]);
  ks_st32B(s + 32, h, A[4], A[5], A[6], A[7]);
}

template <class P>
__global__ void __launch_bounds__(SPLIT_WG) k2_kp_post(int n, const uint8_t* __restrict__ coins, const uint8_t* __restrict__ pk,
                                                       uint8_t* __restrict__ sk) {
  KS_PROLOGUE(n);
  uint32_t A[25], z[4];
  ks_absorb_words<136, 0x06, P::PKB>(A, pk + (size_t)i * P::PKB, h);
  uint8_t* tail = sk + (size_t)i * P::SKB + 2 * P::PVB + 32;
  ks_st32B(tail, h, A[0], A[1], A[2], A[3]);
  ks_ld32B(coins + (size_t)i * 64 + 32, h, z);
  ks_st32B(tail + 32, h, z[0], z[1], z[2], z[3]);
}

template <class P>
__global__ void __launch_bounds__(SPLIT_WG) k2_hash_pk(const uint8_t* __restrict__ pk, uint8_t* __restrict__ out) {
  KS_PROLOGUE(1);
  uint32_t A[25];
  ks_absorb_words<136, 0x06, P::PKB>(A, pk, h);
  ks_st32B(out, h, A[0], A[1], A[2], A[3]);
}

// encaps, before (k_enc_pre of kem_kernels.h) for work-item g of the pairs: instance g >> 1, half g & 1
template <class P>
__device__ __forceinline__ void ks_enc_pre_body(int g, int n, const uint8_t* __restrict__ coins, const uint8_t* __restrict__ pk,
                                                const uint8_t* __restrict__ hpk, uint8_t* __restrict__ small) {
  const int i = g >> 1;
  if (i >= n) return;
  const KsHalf h = ks_half(g & 1);
  uint32_t A[25], m[4], hp[4];
  ks_ld32B(coins + (size_t)i * 32, h, m);
  ks_sha3_256_32(A, m, h);
#pragma unroll
  for (int w = 0; w < 4; w++) m[w] = A[w];
  if (hpk != nullptr) {
    ks_ld32B(hpk, h, hp);
  } else {
    ks_absorb_words<136, 0x06, P::PKB>(A, pk + (size_t)i * P::PKB, h);
#pragma unroll
    for (int w = 0; w < 4; w++) hp[w] = A[w];
  }
  ks_sha3_512_64(A, m, hp, h);
  uint8_t* s = small + (size_t)i * SMALL;
  ks_st32B(s, h, m[0], m[1], m[2], m[3]);
  ks_st32B(s + 32, h, A[0], A[1], A[2], A[3]);
  ks_st32B(s + 64, h, A[4], A[5], A[6], A[7]);
}
template <class P>
__global__ void __launch_bounds__(SPLIT_WG) k2_enc_pre(int n, const uint8_t* __restrict__ coins, const uint8_t* __restrict__ pk,
                                                       const uint8_t* __restrict__ hpk, uint8_t* __restrict__ small) {
  ks_enc_pre_body<P>(blockIdx.x * SPLIT_WG + threadIdx.x, n, coins, pk, hpk, small);
}

// ------------------------------------------------------------------ front kernels of small batches
// Up to FRONT_MAX_N instances no kernel fills the chip for long and a call is the sum of its dependent launches, each as
// long as its slowest lane.  Two pairs of neighbours in those chains do not depend on each other -- in an
// encapsulation the hashes in front (H(m), H(pk), G: eleven permutations) and the matrix expansion (three, from rho
// alone); in a key generation the matrix expansion and the PRF (both from G(d)) -- so each pair is ONE launch whose
// first workgroups do the one job and the rest the other, every workgroup a single wavefront.  They run side by
// side on the idle chip: the shorter job and one launch gap disappear from the chain, without a second stream.
// Threshold by same-box A/B on device-pointer calls (profiles/r03_ab_latency_front_kernels.txt): keypair / enc at n = 1
// 152 / 219 -> 134 / 183 us, 8192: 195 / 252 -> 178 / 198, 16384: 262 / 310 -> 215 / 243, 65536: 605 / 688 -> 573 / 643,
// 131072 a tie, 262144 +1 / +4 %: there the matrix kernel wants its 512-lane workgroups and phase lock back (keccak.h).
#ifndef KY_FRONT_MAX_N
#define KY_FRONT_MAX_N 65536
#endif
constexpr size_t FRONT_MAX_N = KY_FRONT_MAX_N;
template <class P>
__global__ void __launch_bounds__(SPLIT_WG, 4) k_enc_front(int n, int nb_pre, const uint8_t* __restrict__ coins, const uint8_t* __restrict__ pk,
                                                           size_t pk_stride_rho, uint8_t* __restrict__ small, int16_t* __restrict__ mat,
                                                           uint32_t* __restrict__ tailq, uint32_t tail_cap) {
  static_assert(SPLIT_WG == SPONGE_WG_SMALL, "one wavefront per workgroup in both jobs");
  if ((int)blockIdx.x < nb_pre)
    ks_enc_pre_body<P>(blockIdx.x * SPLIT_WG + threadIdx.x, n, coins, pk, nullptr, small);
  else  // A^T from the rho of every public key (ref/indcpa.c:270)
    sample_matrix_body<P, SPONGE_WG_SMALL, 0>(blockIdx.x - nb_pre, n * P::NMAT, pk + P::PVB, pk_stride_rho, 1, mat, tailq, tail_cap);
}
template <class P>
__global__ void __launch_bounds__(SPLIT_WG, 4) k_kp_front(int n, int nb_mat, uint8_t* __restrict__ small, int16_t* __restrict__ mat,
                                                          uint32_t* __restrict__ tailq, uint32_t tail_cap, uint8_t* __restrict__ prf) {
  if ((int)blockIdx.x < nb_mat)  // A from rho = G(d)[0..31] (ref/indcpa.c:219)
    sample_matrix_body<P, SPONGE_WG_SMALL, 0>(blockIdx.x, n * P::NMAT, small, SMALL, 0, mat, tailq, tail_cap);
  else  // s, e from sigma = G(d)[32..63], nonces 0..2K-1 (ref/indcpa.c:220-223)
    prf_body<P>((long)(blockIdx.x - nb_mat) * SPLIT_WG + threadIdx.x, n, 2 * P::K, 2 * P::K, small, 32, prf);
}

// decaps, middle (k_dec_mid of kem_kernels.h) for work-item g of the pairs
template <class P>
__device__ __forceinline__ void ks_dec_mid_body(int g, int n, const uint8_t* __restrict__ sk, size_t sk_stride, uint8_t* __restrict__ small) {
  const int i = g >> 1;
  if (i >= n) return;
  const KsHalf h = ks_half(g & 1);
  uint32_t A[25], m[4], hp[4];
  uint8_t* s = small + (size_t)i * SMALL;
  ks_ld32B(s, h, m);
  ks_ld32B(sk + (size_t)i * sk_stride + 2 * P::PVB + 32, h, hp);
  ks_sha3_512_64(A, m, hp, h);
  ks_st32B(s + 32, h, A[0], A[1], A[2], A[3]);
  ks_st32B(s + 64, h, A[4], A[5], A[6], A[7]);
}
template <class P>
__global__ void __launch_bounds__(SPLIT_WG) k2_dec_mid(int n, const uint8_t* __restrict__ sk, size_t sk_stride, uint8_t* __restrict__ small) {
  ks_dec_mid_body<P>(blockIdx.x * SPLIT_WG + threadIdx.x, n, sk, sk_stride, small);
}

template <class P, bool DEC>
__global__ void __launch_bounds__(SPLIT_WG) k2_post(int n, const uint8_t* __restrict__ ct, const uint8_t* __restrict__ small,
                                                    const uint8_t* __restrict__ sk, size_t sk_stride, uint8_t* __restrict__ ss) {
  KS_PROLOGUE(n);
  uint32_t A[25], kb[4], hc[4];
  ks_absorb_words<136, 0x06, P::CTB>(A, ct + (size_t)i * P::CTB, h);
#pragma unroll
  for (int w = 0; w < 4; w++) hc[w] = A[w];
  const uint8_t* s = small + (size_t)i * SMALL;
  ks_ld32B(s + 32, h, kb);
  if constexpr (DEC) {  // constant-time select as ref/verify.c:40-47
    uint32_t z[4];
    ks_ld32B(sk + (size_t)i * sk_stride + 2 * P::PVB + 64, h, z);
    const uint32_t mask = 0u - (*reinterpret_cast<const uint32_t*>(s + 96) != 0u ? 1u : 0u);
#pragma unroll
    for (int w = 0; w < 4; w++) kb[w] ^= mask & (kb[w] ^ z[w]);
  }
  ks_shake256_64(A, kb, hc, h);
  ks_st32B(ss + (size_t)i * 32, h, A[0], A[1], A[2], A[3]);
}
#undef KS_PROLOGUE

// ------------------------------------------------------------------ the middle launch of a small decapsulation
// ss = KDF(Kbar'-or-z || H(ct)) (ref/kem.c:119-125): the nine permutations of H(ct) need nothing but the ciphertext the
// caller handed in, yet as part of k_post they sit at the END of the chain, behind the re-encryption.  For batches up to
// FRONT_MAX_N they move forward into ONE launch with the two other jobs that can start once m' is known and do not
// depend on each other: the matrix expansion (three permutations from the key's rho) and G(m' || H(pk)) (one).  Its first
// workgroups sample, the next hash ciphertexts on pairs, the last compute G on pairs; H(ct) waits in the caller's ss
// buffer, and what is left behind the re-encryption is the one permutation of the KDF (k2_kdf_dec).
template <class P>
__device__ __forceinline__ void ks_hash_ct_body(int g, int n, const uint8_t* __restrict__ ct, uint8_t* __restrict__ out) {
  const int i = g >> 1;
  if (i >= n) return;
  const KsHalf h = ks_half(g & 1);
  uint32_t A[25];
  ks_absorb_words<136, 0x06, P::CTB>(A, ct + (size_t)i * P::CTB, h);
  ks_st32B(out + (size_t)i * 32, h, A[0], A[1], A[2], A[3]);
}
template <class P>
__global__ void __launch_bounds__(SPLIT_WG, 4) k_dec_middle(int n, int nb_mat, int nb_h, const uint8_t* __restrict__ sk, int16_t* __restrict__ mat,
                                                            uint32_t* __restrict__ tailq, uint32_t tail_cap, const uint8_t* __restrict__ ct,
                                                            uint8_t* __restrict__ ss, uint8_t* __restrict__ small) {
  const int b = (int)blockIdx.x;
  if (b < nb_mat)  // A^T from the rho inside the secret key's copy of pk (ref/indcpa.c:270 via ref/kem.c:114)
    sample_matrix_body<P, SPONGE_WG_SMALL, 0>(blockIdx.x, n * P::NMAT, sk + 2 * P::PVB, P::SKB, 1, mat, tailq, tail_cap);
  else if (b < nb_mat + nb_h)
    ks_hash_ct_body<P>((b - nb_mat) * SPLIT_WG + threadIdx.x, n, ct, ss);
  else  // (Kbar', r') = G(m' || H(pk)), ref/kem.c:109-111
    ks_dec_mid_body<P>((b - nb_mat - nb_h) * SPLIT_WG + threadIdx.x, n, sk, P::SKB, small);
}
// the same hash on its own (one key for the whole batch: no matrix to expand beside it)
template <class P>
__global__ void __launch_bounds__(SPLIT_WG) k2_hash_ct(int n, const uint8_t* __restrict__ ct, uint8_t* __restrict__ ss) {
  ks_hash_ct_body<P>(blockIdx.x * SPLIT_WG + threadIdx.x, n, ct, ss);
}
// ss holds H(ct) on entry and the shared secret on return; constant-time select as ref/verify.c:40-47
template <class P>
__global__ void __launch_bounds__(SPLIT_WG) k2_kdf_dec(int n, const uint8_t* __restrict__ small, const uint8_t* __restrict__ sk,
                                                       size_t sk_stride, uint8_t* __restrict__ ss) {
  const int g = blockIdx.x * SPLIT_WG + threadIdx.x, i = g >> 1;
  if (i >= n) return;
  const KsHalf h = ks_half(g & 1);
  uint32_t A[25], kb[4], hc[4], z[4];
  const uint8_t* s = small + (size_t)i * SMALL;
  ks_ld32B(ss + (size_t)i * 32, h, hc);
  ks_ld32B(s + 32, h, kb);
  ks_ld32B(sk + (size_t)i * sk_stride + 2 * P::PVB + 64, h, z);
  const uint32_t mask = 0u - (*reinterpret_cast<const uint32_t*>(s + 96) != 0u ? 1u : 0u);
#pragma unroll
  for (int w = 0; w < 4; w++) kb[w] ^= mask & (kb[w] ^ z[w]);
  ks_shake256_64(A, kb, hc, h);
  ks_st32B(ss + (size_t)i * 32, h, A[0], A[1], A[2], A[3]);
}

// ------------------------------------------------------------------ ML-KEM (FIPS 203) on the same routes
// The FO glue of kem_kernels.h's ML-KEM kernels (k_ml_kp_pre, k_ml_enc_pre, k_ml_post_dec) on pairs, and the two
// one-launch stages of a small batch: what the standard changed moves MORE work off the end of the chain than
// round 3 allowed -- an encapsulation ends with the ciphertext (the shared secret is G's first half, nothing is
// hashed behind the encryption), and a decapsulation's rejection key J(z || c) needs only inputs the caller handed
// in, so it runs beside the matrix expansion and G, and what is left behind the re-encryption is a select.

// keypair, before: (rho, sigma) = G(d || k)
__global__ void __launch_bounds__(SPLIT_WG) k2_ml_kp_pre(int n, int k, const uint8_t* __restrict__ coins, uint8_t* __restrict__ small) {
  const int g = blockIdx.x * SPLIT_WG + threadIdx.x, i = g >> 1;
  if (i >= n) return;
  const KsHalf h = ks_half(g & 1);
  uint32_t d[4], A[25];
  ks_ld32B(coins + (size_t)i * 64, h, d);
  ks_zero(A);
#pragma unroll
  for (int w = 0; w < 4; w++) A[w] = d[w];
  A[4] = h.low((uint32_t)k | 0x0600u);  // byte 32 = k, then the SHA3 domain byte
  A[8] = h.high(0x80000000u);
  ks_f1600(A, h);
  uint8_t* s = small + (size_t)i * SMALL;
  ks_st32B(s, h, A[0], A[1], A[2], A[3]);
  ks_st32B(s + 32, h, A[4], A[5], A[6], A[7]);
}

// encaps: (K, r) = G(m || H(ek)); the shared secret is K itself
template <class P>
__device__ __forceinline__ void ks_ml_enc_pre_body(int g, int n, const uint8_t* __restrict__ coins, const uint8_t* __restrict__ pk,
                                                   uint8_t* __restrict__ small, uint8_t* __restrict__ ss) {
  const int i = g >> 1;
  if (i >= n) return;
  const KsHalf h = ks_half(g & 1);
  uint32_t A[25], m[4], hp[4];
  ks_ld32B(coins + (size_t)i * 32, h, m);
  ks_absorb_words<136, 0x06, P::PKB>(A, pk + (size_t)i * P::PKB, h);
#pragma unroll
  for (int w = 0; w < 4; w++) hp[w] = A[w];
  ks_sha3_512_64(A, m, hp, h);
  uint8_t* s = small + (size_t)i * SMALL;
  ks_st32B(s, h, m[0], m[1], m[2], m[3]);
  ks_st32B(ss + (size_t)i * 32, h, A[0], A[1], A[2], A[3]);
  ks_st32B(s + 64, h, A[4], A[5], A[6], A[7]);
}
template <class P>
__global__ void __launch_bounds__(SPLIT_WG, 4) k_ml_enc_front(int n, int nb_pre, const uint8_t* __restrict__ coins, const uint8_t* __restrict__ pk,
                                                              size_t pk_stride_rho, uint8_t* __restrict__ small, uint8_t* __restrict__ ss,
                                                              int16_t* __restrict__ mat, uint32_t* __restrict__ tailq, uint32_t tail_cap) {
  if ((int)blockIdx.x < nb_pre)
    ks_ml_enc_pre_body<P>(blockIdx.x * SPLIT_WG + threadIdx.x, n, coins, pk, small, ss);
  else
    sample_matrix_body<P, SPONGE_WG_SMALL, 0>(blockIdx.x - nb_pre, n * P::NMAT, pk + P::PVB, pk_stride_rho, 1, mat, tailq, tail_cap);
}

// decaps: the rejection key J(z || c) = SHAKE256(z || c, 32) into `out`.  Word w of the message is z's word w for
// w < 4 and the ciphertext's word w - 4 after it; every block is loaded while the one before it is permuted.
template <class P>
__device__ __forceinline__ void ks_ml_j_body(int g, int n, const uint8_t* __restrict__ ct, const uint8_t* __restrict__ sk, size_t sk_stride,
                                             uint8_t* __restrict__ out) {
  const int i = g >> 1;
  if (i >= n) return;
  const KsHalf h = ks_half(g & 1);
  constexpr int RW = 17, TOTW = 4 + P::CTB / 8, NFULL = TOTW / RW, TAILW = TOTW % RW;
  static_assert(NFULL >= 2, "the first block is peeled");
  const uint8_t* c = ct + (size_t)i * P::CTB;  // word w of the message, w >= 4, lies at c + 8 (w - 4)
  uint32_t A[25], nx[RW];
  ks_zero(A);
#pragma unroll
  for (int w = 0; w < 4; w++) A[w] = ks_ld(sk + (size_t)i * sk_stride + 2 * P::PVB + 64 + 8 * w, h);
#pragma unroll
  for (int w = 4; w < RW; w++) A[w] = ks_ld(c + 8 * (w - 4), h);
#pragma unroll
  for (int w = 0; w < RW; w++) nx[w] = ks_ld(c + 8 * (RW + w - 4), h);
  ks_f1600(A, h);
#pragma unroll 1
  for (int b = 1; b < NFULL; b++) {
#pragma unroll
    for (int w = 0; w < RW; w++) A[w] ^= nx[w];
#pragma unroll
    for (int w = 0; w < RW; w++) {  // the block after this one; past the end the last word again (not absorbed)
      const int idx = (b + 1) * RW + w;
      nx[w] = ks_ld(c + 8 * ((idx < TOTW ? idx : TOTW - 1) - 4), h);
    }
    ks_f1600(A, h);
  }
#pragma unroll
  for (int w = 0; w < TAILW; w++) A[w] ^= nx[w];
  A[TAILW] ^= h.low(0x1Fu);
  A[RW - 1] ^= h.high(0x80000000u);
  ks_f1600(A, h);
  ks_st32B(out + (size_t)i * 32, h, A[0], A[1], A[2], A[3]);
}
// the middle launch of a small decapsulation: matrix expansion | J(z || c) on pairs | G(m' || h) on pairs
template <class P>
__global__ void __launch_bounds__(SPLIT_WG, 4) k_ml_dec_middle(int n, int nb_mat, int nb_h, const uint8_t* __restrict__ sk, int16_t* __restrict__ mat,
                                                               uint32_t* __restrict__ tailq, uint32_t tail_cap, const uint8_t* __restrict__ ct,
                                                               uint8_t* __restrict__ ss, uint8_t* __restrict__ small) {
  const int b = (int)blockIdx.x;
  if (b < nb_mat)
    sample_matrix_body<P, SPONGE_WG_SMALL, 0>(blockIdx.x, n * P::NMAT, sk + 2 * P::PVB, P::SKB, 1, mat, tailq, tail_cap);
  else if (b < nb_mat + nb_h)
    ks_ml_j_body<P>((b - nb_mat) * SPLIT_WG + threadIdx.x, n, ct, sk, P::SKB, ss);
  else
    ks_dec_mid_body<P>((b - nb_mat - nb_h) * SPLIT_WG + threadIdx.x, n, sk, P::SKB, small);
}
// ss holds J(z || c) on entry; ss = (c == c') ? K' : J, selected without a branch.  One work-item per dword.
__global__ void __launch_bounds__(256) k_ml_select(int n, const uint8_t* __restrict__ small, uint8_t* __restrict__ ss) {
  const int t = blockIdx.x * 256 + threadIdx.x, i = t >> 3, w = t & 7;
  if (i >= n) return;
  const uint8_t* s = small + (size_t)i * SMALL;
  const uint32_t mask = 0u - (*reinterpret_cast<const uint32_t*>(s + 96) != 0u ? 1u : 0u);
  const uint32_t kb = *reinterpret_cast<const uint32_t*>(s + 32 + 4 * w);
  uint32_t* o = reinterpret_cast<uint32_t*>(ss + (size_t)i * 32) + w;
  *o = kb ^ (mask & (kb ^ *o));
}

}  // namespace ky
