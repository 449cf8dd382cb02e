// kem_kernels.h -- the batched KEM as a short pipeline of gfx950 kernels.
//
// Two work decompositions alternate, each the natural one for its stage:
//
//   sponge layout  one work-item per Keccak sponge (64 sponges per wavefront,
//                  state in VGPRs): FO hashes (one sponge per KEM instance),
//                  SHAKE128 matrix expansion with rejection sampling (one per
//                  matrix entry) and the SHAKE256 PRF (one per noise poly).
//   poly layout    32 work-items per polynomial, 8 instances per 256-thread
//                  workgroup, polynomials in LDS: NTT, base multiplication,
//                  inverse NTT, noise, compression and packing.
//
// Between stages the per-instance intermediates (expanded matrix, PRF bytes,
// 128 bytes of seeds/flags) live in a device workspace laid out so that both
// producer and consumer touch it with contiguous, 16-byte-aligned accesses.
//
// Reference call trees restated: ref/kem.c:23-127, ref/indcpa.c:164-329.
#pragma once
#include "keccak.h"
#include "field.h"
#include "params.h"

namespace ky {

// ------------------------------------------------------------------ small hashes
// sha3_256 of 32 bytes already in registers
__device__ __forceinline__ void sha3_256_32(u64p (&A)[25], const u64p (&in)[4]) {
  keccak_zero(A);
#pragma unroll
  for (int w = 0; w < 4; w++) A[w] = in[w];
  A[4].lo = 0x06;
  A[16].hi = 0x80000000u;
  keccak_f1600(A);
}
// sha3_512 of 64 bytes (two 32-byte halves in registers) -> A[0..7]
__device__ __forceinline__ void sha3_512_64(u64p (&A)[25], const u64p (&a)[4], const u64p (&b)[4]) {
  keccak_zero(A);
#pragma unroll
  for (int w = 0; w < 4; w++) { A[w] = a[w]; A[4 + w] = b[w]; }
  A[8].lo = 0x06;
  A[8].hi = 0x80000000u;
  keccak_f1600(A);
}
// shake256(64 bytes) -> first 32 bytes in A[0..3]           (kdf, ref/symmetric.h:56)
__device__ __forceinline__ void shake256_64(u64p (&A)[25], const u64p (&a)[4], const u64p (&b)[4]) {
  keccak_zero(A);
#pragma unroll
  for (int w = 0; w < 4; w++) { A[w] = a[w]; A[4 + w] = b[w]; }
  A[8].lo = 0x1F;
  A[16].hi = 0x80000000u;
  keccak_f1600(A);
}

__device__ __forceinline__ void ld32B(const uint8_t* p, u64p (&v)[4]) {
  uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 16);
  v[0] = u64p{a.x, a.y}; v[1] = u64p{a.z, a.w}; v[2] = u64p{b.x, b.y}; v[3] = u64p{b.z, b.w};
}
__device__ __forceinline__ void st32B(uint8_t* p, const u64p& a, const u64p& b, const u64p& c, const u64p& d) {
  *reinterpret_cast<uint4*>(p) = make_uint4(a.lo, a.hi, b.lo, b.hi);
  *reinterpret_cast<uint4*>(p + 16) = make_uint4(c.lo, c.hi, d.lo, d.hi);
}

// ------------------------------------------------------------------ FO hashes
// keypair, before: (rho, sigma) = G(d).                        ref/indcpa.c:215-216
__global__ void __launch_bounds__(SPONGE_WG, 4) k_kp_pre(int n, const uint8_t* __restrict__ coins, uint8_t* __restrict__ small) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64p d[4], A[25];
  ld32B(coins + (size_t)i * 64, d);
  keccak_zero(A);
#pragma unroll
  for (int w = 0; w < 4; w++) A[w] = d[w];
  A[4].lo = 0x06;
  A[8].hi = 0x80000000u;
  keccak_f1600(A);
  uint8_t* s = small + (size_t)i * SMALL;
  st32B(s, A[0], A[1], A[2], A[3]);
  st32B(s + 32, A[4], A[5], A[6], A[7]);
}

// keypair, after: sk tail = H(pk) || z.                          ref/kem.c:30-32
template <class P>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_kp_post(int n, const uint8_t* __restrict__ coins, const uint8_t* __restrict__ pk,
                                                 uint8_t* __restrict__ sk) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64p A[25], z[4];
  keccak_absorb_words<136, 0x06, P::PKB>(A, pk + (size_t)i * P::PKB);
  uint8_t* tail = sk + (size_t)i * P::SKB + 2 * P::PVB + 32;
  st32B(tail, A[0], A[1], A[2], A[3]);
  ld32B(coins + (size_t)i * 64 + 32, z);
  st32B(tail + 32, z[0], z[1], z[2], z[3]);
}

// H(pk) of one key, for the shared-key entry points.
template <class P>
__global__ void __launch_bounds__(64) k_hash_pk(const uint8_t* __restrict__ pk, uint8_t* __restrict__ out) {
  if (threadIdx.x | blockIdx.x) return;
  u64p A[25];
  keccak_absorb_words<136, 0x06, P::PKB>(A, pk);
  st32B(out, A[0], A[1], A[2], A[3]);
}

// encaps, before: m = H(m0); (Kbar, r) = G(m || H(pk)).         ref/kem.c:59-65
// hpk != null: every instance uses that one precomputed H(pk) (shared key).
template <class P>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_enc_pre(int n, const uint8_t* __restrict__ coins, const uint8_t* __restrict__ pk,
                                                 const uint8_t* __restrict__ hpk, uint8_t* __restrict__ small) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64p A[25], m[4], h[4];
  ld32B(coins + (size_t)i * 32, m);
  sha3_256_32(A, m);
#pragma unroll
  for (int w = 0; w < 4; w++) m[w] = A[w];
  if (hpk != nullptr) {
    ld32B(hpk, h);
  } else {
    keccak_absorb_words<136, 0x06, P::PKB>(A, pk + (size_t)i * P::PKB);
#pragma unroll
    for (int w = 0; w < 4; w++) h[w] = A[w];
  }
  sha3_512_64(A, m, h);
  uint8_t* s = small + (size_t)i * SMALL;
  st32B(s, m[0], m[1], m[2], m[3]);
  st32B(s + 32, A[0], A[1], A[2], A[3]);
  st32B(s + 64, A[4], A[5], A[6], A[7]);
}

// decaps, middle: (Kbar', r') = G(m' || H(pk) from sk).          ref/kem.c:109-111
template <class P>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_dec_mid(int n, const uint8_t* __restrict__ sk, size_t sk_stride,
                                                 uint8_t* __restrict__ small) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64p A[25], m[4], h[4];
  uint8_t* s = small + (size_t)i * SMALL;
  ld32B(s, m);
  ld32B(sk + (size_t)i * sk_stride + 2 * P::PVB + 32, h);
  sha3_512_64(A, m, h);
  st32B(s + 32, A[0], A[1], A[2], A[3]);
  st32B(s + 64, A[4], A[5], A[6], A[7]);
}

// encaps/decaps, after: ss = KDF(Kbar-or-z || H(ct)).            ref/kem.c:71-73, :116-125
template <class P, bool DEC>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_post(int n, const uint8_t* __restrict__ ct, const uint8_t* __restrict__ small,
                                              const uint8_t* __restrict__ sk, size_t sk_stride, uint8_t* __restrict__ ss) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64p A[25], kb[4], h[4];
  keccak_absorb_words<136, 0x06, P::CTB>(A, ct + (size_t)i * P::CTB);
#pragma unroll
  for (int w = 0; w < 4; w++) h[w] = A[w];
  const uint8_t* s = small + (size_t)i * SMALL;
  ld32B(s + 32, kb);
  if constexpr (DEC) {
    // constant-time select as ref/verify.c:40-47: kb ^= mask & (kb ^ z)
    u64p z[4];
    ld32B(sk + (size_t)i * sk_stride + 2 * P::PVB + 64, z);
    uint32_t mask = 0u - (*reinterpret_cast<const uint32_t*>(s + 96) != 0u ? 1u : 0u);
#pragma unroll
    for (int w = 0; w < 4; w++) {
      kb[w].lo ^= mask & (kb[w].lo ^ z[w].lo);
      kb[w].hi ^= mask & (kb[w].hi ^ z[w].hi);
    }
  }
  shake256_64(A, kb, h);
  st32B(ss + (size_t)i * 32, A[0], A[1], A[2], A[3]);
}

// ------------------------------------------------------------------ ML-KEM (FIPS 203) FO glue
// Not in the reference: the same K-PKE underneath -- matrix, noise,
// transforms and codecs are bit for bit those of round-3 Kyber -- with the final standard's
// key derivation around it.  FIPS 203 algorithms 13 (K-PKE.KeyGen, G(d || k)), 17
// (Encaps_internal: (K, r) = G(m || H(ek)), no hashing of m, no KDF) and 18
// (Decaps_internal: implicit rejection key J(z || c) = SHAKE256(z || c, 32)).

// keypair, before: (rho, sigma) = G(d || k)
__global__ void __launch_bounds__(SPONGE_WG, 4) k_ml_kp_pre(int n, int k, const uint8_t* __restrict__ coins, uint8_t* __restrict__ small) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64p d[4], A[25];
  ld32B(coins + (size_t)i * 64, d);
  keccak_zero(A);
#pragma unroll
  for (int w = 0; w < 4; w++) A[w] = d[w];
  A[4].lo = (uint32_t)k | 0x0600u;  // byte 32 = k, then the SHA3 domain byte
  A[8].hi = 0x80000000u;
  keccak_f1600(A);
  uint8_t* s = small + (size_t)i * SMALL;
  st32B(s, A[0], A[1], A[2], A[3]);
  st32B(s + 32, A[4], A[5], A[6], A[7]);
}

// encaps: (K, r) = G(m || H(ek)); the shared secret is K itself
template <class P>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_ml_enc_pre(int n, const uint8_t* __restrict__ coins, const uint8_t* __restrict__ pk,
                                                    uint8_t* __restrict__ small, uint8_t* __restrict__ ss) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64p A[25], m[4], h[4];
  ld32B(coins + (size_t)i * 32, m);
  keccak_absorb_words<136, 0x06, P::PKB>(A, pk + (size_t)i * P::PKB);
#pragma unroll
  for (int w = 0; w < 4; w++) h[w] = A[w];
  sha3_512_64(A, m, h);
  uint8_t* s = small + (size_t)i * SMALL;
  st32B(s, m[0], m[1], m[2], m[3]);
  st32B(ss + (size_t)i * 32, A[0], A[1], A[2], A[3]);
  st32B(s + 64, A[4], A[5], A[6], A[7]);
}

// decaps, after: ss = (c == c') ? K' : J(z || c), selected without a branch
template <class P>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_ml_post_dec(int n, const uint8_t* __restrict__ ct, const uint8_t* __restrict__ small,
                                                     const uint8_t* __restrict__ sk, size_t sk_stride, uint8_t* __restrict__ ss) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int RW = 17, TOTW = 4 + P::CTB / 8, NFULL = TOTW / RW, TAILW = TOTW % RW;
  const uint8_t* c = ct + (size_t)i * P::CTB;
  u64p A[25], z[4], kb[4];
  ld32B(sk + (size_t)i * sk_stride + 2 * P::PVB + 64, z);
  keccak_zero(A);
  // block 0: z, then the first 13 words of c; every later block is 17 consecutive words of c
#pragma unroll
  for (int w = 0; w < 4; w++) A[w] = z[w];
#pragma unroll
  for (int w = 4; w < RW; w++) A[w] ^= ld64(c + 8 * (w - 4));
  keccak_f1600(A);
#pragma unroll 1
  for (int b = 1; b < NFULL; b++) {
#pragma unroll
    for (int w = 0; w < RW; w++) A[w] ^= ld64(c + 8 * (b * RW + w - 4));
    keccak_f1600(A);
  }
#pragma unroll
  for (int w = 0; w < TAILW; w++) A[w] ^= ld64(c + 8 * (NFULL * RW + w - 4));
  A[TAILW].lo ^= 0x1Fu;
  A[RW - 1].hi ^= 0x80000000u;
  keccak_f1600(A);
  const uint8_t* s = small + (size_t)i * SMALL;
  ld32B(s + 32, kb);
  const uint32_t mask = 0u - (*reinterpret_cast<const uint32_t*>(s + 96) != 0u ? 1u : 0u);
  u64p o[4];
#pragma unroll
  for (int w = 0; w < 4; w++) {
    o[w].lo = kb[w].lo ^ (mask & (kb[w].lo ^ A[w].lo));
    o[w].hi = kb[w].hi ^ (mask & (kb[w].hi ^ A[w].hi));
  }
  st32B(ss + (size_t)i * 32, o[0], o[1], o[2], o[3]);
}

// Input checks of FIPS 203 7.2 / 7.3 as per-instance flags (0 = passes):
//   ek: every 12-bit coefficient of the encoded t^ is below q ("modulus check")
//   dk: the same for the embedded ek, and H(ek) equals the stored hash ("hash check")
template <class P, bool DK>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_ml_check(int n, const uint8_t* __restrict__ key, uint32_t* __restrict__ flags) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* ek = key + (size_t)i * (DK ? P::SKB : P::PKB) + (DK ? P::PVB : 0);
  uint32_t bad = 0;
#pragma unroll 1
  for (int w = 0; w < P::PVB / 12; w++) {  // three dwords = eight coefficients
    const uint32_t* p = reinterpret_cast<const uint32_t*>(ek) + 3 * w;
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
    const uint32_t c[8] = {w0 & 0xFFF, (w0 >> 12) & 0xFFF, (w0 >> 24 | w1 << 8) & 0xFFF, (w1 >> 4) & 0xFFF,
                           (w1 >> 16) & 0xFFF, (w1 >> 28 | w2 << 4) & 0xFFF, (w2 >> 8) & 0xFFF, w2 >> 20};
#pragma unroll
    for (int k = 0; k < 8; k++) bad |= (uint32_t)(Q - 1 - (int)c[k]) >> 31;
  }
  if constexpr (DK) {
    u64p A[25], h[4];
    keccak_absorb_words<136, 0x06, P::PKB>(A, ek);
    ld32B(ek + P::PKB, h);
    uint32_t d = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) d |= (A[w].lo ^ h[w].lo) | (A[w].hi ^ h[w].hi);
    bad |= ((d | (0u - d)) >> 31) << 1;
  }
  flags[i] = bad;
}

// ------------------------------------------------------------------ SHAKE256 PRF
// One work-item per (nonce, instance); jobs are nonce-major so a wavefront is
// uniform in the output length.  Writes the raw PRF bytes; the centred
// binomial map is applied by the consumer in poly layout.
// ref/symmetric-shake.c:43-51 + ref/poly.c:208-232 (prf part).
// (a device function: k_prf runs it on its own, k_kp_front beside the matrix expansion)
template <class P>
__device__ __forceinline__ void prf_body(long gid, int n, int njobs_per_inst, int n_eta1, const uint8_t* __restrict__ small,
                                         int seed_off, uint8_t* __restrict__ prf) {
  if (gid >= (long)n * njobs_per_inst) return;
  int nonce = (int)(gid / n), inst = (int)(gid % n);
  u64p A[25], key[4];
  ld32B(small + (size_t)inst * SMALL + seed_off, key);
  keccak_zero(A);
#pragma unroll
  for (int w = 0; w < 4; w++) A[w] = key[w];
  A[4].lo = (uint32_t)nonce | 0x1F00u;
  A[16].hi = 0x80000000u;
  keccak_f1600(A);
  uint8_t* out = prf + ((size_t)inst * njobs_per_inst + nonce) * P::PRF_STRIDE;
#pragma unroll
  for (int w = 0; w < 16; w += 2) *reinterpret_cast<uint4*>(out + 8 * w) = make_uint4(A[w].lo, A[w].hi, A[w + 1].lo, A[w + 1].hi);
  if constexpr (P::ETA1 == 3) {
    if (nonce < n_eta1) {  // 192 bytes: 17 words of block one, 7 of block two (ref/fips202.c:729-733)
      u64p w16 = A[16];
      keccak_f1600(A);
      *reinterpret_cast<uint4*>(out + 128) = make_uint4(w16.lo, w16.hi, A[0].lo, A[0].hi);
#pragma unroll
      for (int w = 1; w < 7; w += 2)
        *reinterpret_cast<uint4*>(out + 136 + 8 * (w - 1) + 8) = make_uint4(A[w].lo, A[w].hi, A[w + 1].lo, A[w + 1].hi);
    }
  }
}
template <class P>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_prf(int n, int njobs_per_inst, int n_eta1, const uint8_t* __restrict__ small,
                                             int seed_off, uint8_t* __restrict__ prf) {
  prf_body<P>((long)blockIdx.x * blockDim.x + threadIdx.x, n, njobs_per_inst, n_eta1, small, seed_off, prf);
}

// ------------------------------------------------------------------ matrix expansion
// One work-item per matrix entry: SHAKE128(rho || x || y), squeeze 168-byte
// blocks, parse 3-byte groups into two 12-bit candidates, keep those < q until
// 256 are found (ref/indcpa.c:124-145, :164-192; the reference's block
// refill-and-carry is the same as this continuous stream parse because
// 168 % 3 == 0).
//
// Each lane appends its accepted values to a private row of an LDS staging
// tile ("always store, advance only on accept": no divergent branch) and then
// copies whole 64-byte lines of the row to its polynomial with 16-byte stores.  Entries
// that still miss coefficients after three blocks (P ~ 0.9 % each) are handed
// to k_sample_matrix_tail through a queue instead of keeping their wavefront
// squeezing.

template <int BITOFF, int N>
__device__ __forceinline__ uint32_t field12(const uint32_t (&H)[N]) {
  constexpr int k = BITOFF >> 5, s = BITOFF & 31;
  if constexpr (s == 0) return H[k] & 0xFFFu;
  else if constexpr (s == 20) return H[k] >> 20;
  else if constexpr (s < 20) return __builtin_amdgcn_ubfe(H[k], s, 12);
  else return __builtin_amdgcn_alignbit(H[k + 1], H[k], s) & 0xFFFu;
}

// Candidate d is kept iff d < q.  (d - q) is negative exactly then, so bit 1 of
// (d - q) >> 30 is the byte step of the 16-bit write cursor: four full-rate
// VALU ops per candidate, no compare/select, no SGPR traffic.
__device__ __forceinline__ uint32_t accept_step(uint32_t d) { return ((d - (uint32_t)Q) >> 30) & 2u; }

// LDS addressed by its 32-bit byte offset, so the write cursor IS the ds_write address.
typedef __attribute__((address_space(3))) uint16_t lds_u16;
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void lds_st16(uint32_t addr, uint32_t v) { *(lds_u16*)(uintptr_t)addr = (uint16_t)v; }

// FIXED (tools/microbench only): store at a cursor-independent address, to price the bank
// conflicts that diverging cursors cause
template <int G, int N, bool FIXED = false>
__device__ __forceinline__ void parse_group(const uint32_t (&H)[N], uint32_t& waddr, uint32_t fixed = 0) {
  uint32_t d1 = field12<24 * G, N>(H), d2 = field12<24 * G + 12, N>(H);
  if (FIXED && fixed == 0) asm volatile("" ::"v"(d1), "v"(d2));  // no staging stores at all
  else lds_st16(FIXED ? fixed + 4 * (G % 28) : waddr, d1);
  waddr += accept_step(d1);
  if (!(FIXED && fixed == 0)) lds_st16(FIXED ? fixed + 4 * (G % 28) + 2 : waddr, d2);
  waddr += accept_step(d2);
}
// Groups G0..G1-1 of the block.  The compiler, left alone, turns the straightforward loop into one
// serial dependency chain on three registers (extract -> subtract -> shift -> mask -> cursor add ->
// next extract ...): every instruction then waits for its predecessor, 8 cycles on gfx950
// (tools/microbench/valu_rate.hip "dist 1"), and a dependency at distance two is the most expensive
// case of all.  So the candidates and their cursor steps are computed first, as independent
// instructions, BATCH groups at a time, and only the cursor additions form a chain.
constexpr int PARSE_BATCH = 7;
template <int G, int G1, int N>
__device__ __forceinline__ void parse_extract(const uint32_t (&H)[N], uint32_t* d) {
  if constexpr (G < G1) {
    d[0] = field12<24 * G, N>(H);
    d[1] = field12<24 * G + 12, N>(H);
    parse_extract<G + 1, G1, N>(H, d + 2);
  }
}
template <int G0, int G1, int N, bool FIXED = false>
__device__ __forceinline__ void parse_range(const uint32_t (&H)[N], uint32_t& waddr, uint32_t fixed = 0) {
  if constexpr (FIXED) {
    if constexpr (G0 < G1) {
      parse_group<G0, N, FIXED>(H, waddr, fixed);
      parse_range<G0 + 1, G1, N, FIXED>(H, waddr, fixed);
    }
  } else if constexpr (G0 < G1) {
    constexpr int GE = G0 + PARSE_BATCH < G1 ? G0 + PARSE_BATCH : G1, NC = 2 * (GE - G0);
    uint32_t d[NC], st[NC];
    parse_extract<G0, GE, N>(H, d);
#pragma unroll
    for (int c = 0; c < NC; c++) st[c] = accept_step(d[c]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NC; c++) {
      lds_st16(waddr, d[c]);
      waddr += st[c];
    }
    __builtin_amdgcn_sched_barrier(0);
    parse_range<GE, G1, N, FIXED>(H, waddr, fixed);
  }
}

// A 168-byte block is parsed and drained in four parts.  The staging row of a lane always STARTS
// at a 64-byte line of its polynomial: after a part's parse a whole line (32 coefficients) leaves,
// if there is one, as aligned 16-byte stores, and the remainder (< 32 coefficients) moves to the
// front of the row for the next part.  Every line of the matrix is written exactly once, whole,
// and the drain needs no head / tail alignment logic.  (Round 1 drained whatever a half-block
// produced, ragged ~90-byte pieces at 4-byte alignment: most lines were written twice, 1.41x write
// traffic.)
// Row = 31 carried + 28 candidates + 1 spill slot = 30 dwords, padded to an odd stride: 31 dwords
// per lane, 62 KB per 512-lane workgroup, two workgroups per CU -- the residency the phase-locked
// permutation wants (keccak.h).  A/B on one box, Kyber768 2^19: this layout 1.77-1.79 ms; two parts
// per block with 32-byte drain units (37-dword rows) 1.81-1.83 ms; the same four parts without the
// phase lock 1.93 ms; round 2's first layout (256 lanes, two parts, 45-dword rows, three
// workgroups per CU, no lock) 1.84-1.87 ms.
constexpr int MAT_PARTS = 4;
constexpr int MAT_GROUPS = 56 / MAT_PARTS;          // 3-byte groups per part
constexpr int MAT_STAGE_DW = 31;
static_assert(2 * MAT_STAGE_DW >= 31 + 2 * MAT_GROUPS + 1 && MAT_STAGE_DW % 2 == 1, "staging row");
static_assert(31 + 2 * MAT_GROUPS < 64, "at most one whole line per part");
static_assert(2 * SPONGE_WG * MAT_STAGE_DW * 4 + 64 <= 160 * 1024, "two workgroups per CU");

// Tail queue: dword 0 = number of records pushed (may exceed the capacity),
// records of 64 dwords from dword 64: sponge state (50), entry index, count.
constexpr int MAT_MAIN_BLOCKS = 3;
constexpr int TAILQ_HDR_DW = 64, TAILQ_REC_DW = 64;
__host__ __device__ constexpr size_t tailq_capacity(size_t entries) { return entries / 16 + 64; }
__host__ __device__ constexpr size_t tailq_bytes(size_t entries) {
  return 4 * (TAILQ_HDR_DW + tailq_capacity(entries) * TAILQ_REC_DW);
}

template <int G>
__device__ __forceinline__ void tail_parse(const uint32_t (&H)[42], int16_t* out, int& ctr) {
  if constexpr (G < 56) {
    const uint32_t d1 = field12<24 * G, 42>(H), d2 = field12<24 * G + 12, 42>(H);
    if (d1 < (uint32_t)Q && ctr < 256) out[ctr++] = (int16_t)d1;
    if (d2 < (uint32_t)Q && ctr < 256) out[ctr++] = (int16_t)d2;
    tail_parse<G + 1>(H, out, ctr);
  }
}

// ABLATE != 0 only in tools/microbench (1: no drain; 2: no parse either; 3: no drain, staging
// stores at cursor-independent, conflict-free addresses; 4: no drain, no staging stores)
// WG = lanes per workgroup: SPONGE_WG, or SPONGE_WG_SMALL for launches that cannot fill the chip
// (keccak.h); the staging tile is sized by it.
// (a device function over a block index: k_sample_matrix runs it on its own, k_enc_front / k_kp_front as some of their
// workgroups beside the hashes / the PRF of small batches)
template <class P, int WG, int ABLATE>
__device__ __forceinline__ void sample_matrix_body(unsigned bid, int njobs, const uint8_t* __restrict__ rho, size_t rho_stride,
                                                   int transposed, int16_t* __restrict__ mat,
                                                   uint32_t* __restrict__ tailq, uint32_t tail_cap) {
#define KY_BID bid
#include "kem_sample_body.inc"
#undef KY_BID
}
template <class P, int WG = SPONGE_WG, int ABLATE = 0>
__global__ void __launch_bounds__(WG, 4) k_sample_matrix(int njobs, const uint8_t* __restrict__ rho, size_t rho_stride,
                                                                   int transposed, int16_t* __restrict__ mat,
                                                                   uint32_t* __restrict__ tailq, uint32_t tail_cap) {
#define KY_BID blockIdx.x
#include "kem_sample_body.inc"
#undef KY_BID
}

// Finishes the entries parked by k_sample_matrix: one work-item per queue
// record, continues the SHAKE128 stream from the saved sponge and appends
// accepted candidates to the polynomial in place (coefficients below the saved
// count are already in memory).  ~1 % of the entries, a handful of
// coefficients each, so the scattered 2-byte stores do not matter.
__global__ void __launch_bounds__(64) k_sample_matrix_tail(const uint32_t* __restrict__ tailq, uint32_t tail_cap,
                                                           int16_t* __restrict__ mat) {
  const uint32_t cnt = min(tailq[0], tail_cap);
  const uint32_t id = blockIdx.x * 64 + threadIdx.x;
  if (id >= cnt) return;
  const uint32_t* r = tailq + TAILQ_HDR_DW + (size_t)id * TAILQ_REC_DW;
  u64p A[25];
#pragma unroll
  for (int w = 0; w < 24; w += 2) {
    const uint4 v = *reinterpret_cast<const uint4*>(r + 2 * w);
    A[w] = u64p{v.x, v.y};
    A[w + 1] = u64p{v.z, v.w};
  }
  const uint4 last = *reinterpret_cast<const uint4*>(r + 48);
  A[24] = u64p{last.x, last.y};
  int16_t* out = mat + (size_t)last.z * 256;
  int ctr = (int)last.w;
  while (ctr < 256) {
    keccak_f1600(A);
    uint32_t H[42];
#pragma unroll
    for (int w = 0; w < 21; w++) { H[2 * w] = A[w].lo; H[2 * w + 1] = A[w].hi; }
    tail_parse<0>(H, out, ctr);
  }
}

}  // namespace ky
