// kem_kernels_90s.h -- the symmetric layer of the -90s parameter sets
// (ref/symmetric.h:17-32): hash_h/kdf = SHA-256, hash_g = SHA-512, XOF and PRF
// = AES-256-CTR.  Only the sponge-layout kernels differ from the SHAKE sets;
// the poly-layout arithmetic kernels of kem_kernels.h consume the same
// workspace (expanded matrix, raw PRF bytes, 128-byte records) unchanged.
#pragma once
#include "kem_kernels.h"
#include "sha2.h"
#include "aes.h"

namespace ky {

__device__ __forceinline__ void st32Bv(uint8_t* p, const u64p (&v)[4]) { st32B(p, v[0], v[1], v[2], v[3]); }

// keypair, before: (rho, sigma) = sha512(d).                    ref/indcpa.c:215-216
__global__ void __launch_bounds__(256) k90_kp_pre(int n, const uint8_t* __restrict__ coins, uint8_t* __restrict__ small) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  u64p d[4], lo[4], hi[4];
  ld32B(coins + (size_t)i * 64, d);
  sha512_small<1>(d, d, lo, hi);
  uint8_t* s = small + (size_t)i * SMALL;
  st32Bv(s, lo);
  st32Bv(s + 32, hi);
}

// keypair, after: sk tail = sha256(pk) || z.                       ref/kem.c:30-32
template <class P>
__global__ void __launch_bounds__(256) k90_kp_post(int n, const uint8_t* __restrict__ coins, const uint8_t* __restrict__ pk,
                                                   uint8_t* __restrict__ sk) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h[8];
  u64p hp[4], z[4];
  sha256_global<P::PKB>(pk + (size_t)i * P::PKB, h);
  sha256_digest_le(h, hp);
  uint8_t* tail = sk + (size_t)i * P::SKB + 2 * P::PVB + 32;
  st32Bv(tail, hp);
  ld32B(coins + (size_t)i * 64 + 32, z);
  st32Bv(tail + 32, z);
}

// sha256(pk) of one key, for the shared-key entry points.
template <class P>
__global__ void __launch_bounds__(64) k90_hash_pk(const uint8_t* __restrict__ pk, uint8_t* __restrict__ out) {
  if (threadIdx.x | blockIdx.x) return;
  uint32_t h[8];
  u64p hp[4];
  sha256_global<P::PKB>(pk, h);
  sha256_digest_le(h, hp);
  st32Bv(out, hp);
}

// encaps, before: m = sha256(m0); (Kbar, r) = sha512(m || sha256(pk)).   ref/kem.c:59-65
// hpk != null: every instance uses that one precomputed sha256(pk) (shared key).
template <class P>
__global__ void __launch_bounds__(256) k90_enc_pre(int n, const uint8_t* __restrict__ coins, const uint8_t* __restrict__ pk,
                                                   const uint8_t* __restrict__ hpk, uint8_t* __restrict__ small) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h[8];
  u64p m0[4], m[4], hp[4], kb[4], r[4];
  ld32B(coins + (size_t)i * 32, m0);
  sha256_32(m0, m);
  if (hpk != nullptr) {
    ld32B(hpk, hp);
  } else {
    sha256_global<P::PKB>(pk + (size_t)i * P::PKB, h);
    sha256_digest_le(h, hp);
  }
  sha512_small<2>(m, hp, kb, r);
  uint8_t* s = small + (size_t)i * SMALL;
  st32Bv(s, m);
  st32Bv(s + 32, kb);
  st32Bv(s + 64, r);
}

// decaps, middle: (Kbar', r') = sha512(m' || H(pk) from sk).       ref/kem.c:109-111
template <class P>
__global__ void __launch_bounds__(256) k90_dec_mid(int n, const uint8_t* __restrict__ sk, size_t sk_stride,
                                                   uint8_t* __restrict__ small) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  u64p m[4], hp[4], kb[4], r[4];
  uint8_t* s = small + (size_t)i * SMALL;
  ld32B(s, m);
  ld32B(sk + (size_t)i * sk_stride + 2 * P::PVB + 32, hp);
  sha512_small<2>(m, hp, kb, r);
  st32Bv(s + 32, kb);
  st32Bv(s + 64, r);
}

// encaps/decaps, after: ss = sha256(Kbar-or-z || sha256(ct)).      ref/kem.c:71-73, :116-125
template <class P, bool DEC>
__global__ void __launch_bounds__(256) k90_post(int n, const uint8_t* __restrict__ ct, const uint8_t* __restrict__ small,
                                                const uint8_t* __restrict__ sk, size_t sk_stride, uint8_t* __restrict__ ss) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h[8];
  u64p hc[4], kb[4], out[4];
  sha256_global<P::CTB>(ct + (size_t)i * P::CTB, h);
  sha256_digest_le(h, hc);
  const uint8_t* s = small + (size_t)i * SMALL;
  ld32B(s + 32, kb);
  if constexpr (DEC) {
    u64p z[4];
    ld32B(sk + (size_t)i * sk_stride + 2 * P::PVB + 64, z);
    uint32_t mask = 0u - (*reinterpret_cast<const uint32_t*>(s + 96) != 0u ? 1u : 0u);
#pragma unroll
    for (int w = 0; w < 4; w++) {
      kb[w].lo ^= mask & (kb[w].lo ^ z[w].lo);
      kb[w].hi ^= mask & (kb[w].hi ^ z[w].hi);
    }
  }
  sha256_64(kb, hc, out);
  st32Bv(ss + (size_t)i * 32, out);
}

// Key-exchange KDF over NP (2 or 3) 32-byte secrets held in separate dense
// arrays: out = kdf(p0 || p1 [|| p2]).  ref/kex.c:17,26,42,52 with
// ref/symmetric.h:32,56 (SHAKE256 / SHA-256).
template <int NP, bool N90>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_kdf_parts(int n, const uint8_t* __restrict__ p0, const uint8_t* __restrict__ p1,
                                                   const uint8_t* __restrict__ p2, uint8_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64p a[4], b[4], c[4], r[4];
  ld32B(p0 + (size_t)i * 32, a);
  ld32B(p1 + (size_t)i * 32, b);
  if constexpr (NP == 3) ld32B(p2 + (size_t)i * 32, c);
  if constexpr (N90) {
    if constexpr (NP == 3) sha256_96(a, b, c, r);
    else sha256_64(a, b, r);
  } else {
    u64p A[25];
    keccak_zero(A);
#pragma unroll
    for (int w = 0; w < 4; w++) {
      A[w] = a[w];
      A[4 + w] = b[w];
      if constexpr (NP == 3) A[8 + w] = c[w];
    }
    A[4 * NP].lo = 0x1F;
    A[16].hi = 0x80000000u;
    keccak_f1600(A);
#pragma unroll
    for (int w = 0; w < 4; w++) r[w] = A[w];
  }
  st32Bv(out + (size_t)i * 32, r);
}

// PRF: AES-256-CTR(key = seed, nonce = {nonce,0,...}) from counter 0.
// ref/symmetric-aes.c:15-20 -> ref/aes256ctr.c:557-563.  Jobs are nonce-major.
template <class P, int WG = AES_WG>
__global__ void __launch_bounds__(WG) k90_prf(int n, int njobs_per_inst, int n_eta1, const uint8_t* __restrict__ small,
                                              int seed_off, uint8_t* __restrict__ prf) {
  KY_AES_TABLE(tab);
  aes_table_init(tab);
  const AesLds L = aes_lds(tab);
  const long gid = (long)blockIdx.x * WG + threadIdx.x;
  if (gid >= (long)n * njobs_per_inst) return;
  int nonce = (int)(gid / n), inst = (int)(gid % n);
  u64p key[4];
  uint32_t rk[60];
  ld32B(small + (size_t)inst * SMALL + seed_off, key);
  aes256_expand(L, key, rk);
  const AesCtr cx = aes256ctr_init(L, rk, (uint32_t)nonce);
  uint8_t* out = prf + ((size_t)inst * njobs_per_inst + nonce) * P::PRF_STRIDE;
  const int nblk = (P::ETA1 == 3 && nonce < n_eta1) ? 12 : 8;
  // 8 or 12 blocks.  (Four blocks per round of stores, so that a work-item fills whole 64-byte lines back to
  // back, changed neither the time nor WRITE_SIZE, 2.9 x the bytes: the kernel is bound by the look-ups.)
#pragma unroll 1
  for (int b = 0; b < nblk; b += AES_NB) {
    uint32_t o[AES_NB][4];
    aes256ctr_blocks<AES_NB>(L, rk, cx, (uint32_t)b, o);
#pragma unroll
    for (int u = 0; u < AES_NB; u++) *reinterpret_cast<uint4*>(out + 16 * (b + u)) = make_uint4(o[u][0], o[u][1], o[u][2], o[u][3]);
  }
}

// Matrix expansion: AES-256-CTR(key = rho, nonce = {x,y,0,...}) read as one
// byte stream, three bytes at a time (ref/indcpa.c:124-145,164-192 with
// ref/symmetric-aes.c:7-13).  The reference squeezes 512 bytes, then 64 at a
// time carrying buflen % 3 bytes over; that is a continuous parse of the
// keystream, reproduced here 48 bytes (three AES blocks, 16 three-byte groups)
// per step so that no group straddles a step.
constexpr int XOF90_BLOCKS = 3;                                 // AES blocks per step
constexpr int XOF90_GROUPS = 16 * XOF90_BLOCKS / 3;             // 3-byte groups per step
constexpr int STAGE90_DW = (2 * XOF90_GROUPS + 2) / 2;          // 1 carry slot + candidates + 1 spill, in dwords

// Lane-private drain: every lane copies its own staging row to its own
// polynomial (rows are private, so no cross-lane traffic and no readlanes).
// Up to three single dwords bring the destination to a 16-byte boundary, the
// rest leaves as aligned 16-byte stores read from the row at the matching
// offset.  An aligned store may carry up to three dwords past the lane's
// count; they lie inside the same polynomial (whose end is 16-byte aligned) and
// are overwritten by the next piece or by the tail kernel.
// (Rows laid out one per LDS bank, so that the diverging write cursors of the parse never conflict, were
// measured: SQ_LDS_BANK_CONFLICT gone, the kernel's time unchanged, the parse 20 % longer by its address
// arithmetic.  The rows stay consecutive.)
template <int SDW>
__device__ __forceinline__ int drain_private(uint32_t* row, long job, int c0, int c1, uint32_t* mat32) {
  const int d0 = c0 >> 1;
  const int nd = c0 < 256 ? (((c1 < 256 ? c1 : 256) + 1) >> 1) - d0 : 0;
  const int head = min((4 - (d0 & 3)) & 3, nd);
  uint32_t* dst = mat32 + job * 128 + d0;
#pragma unroll
  for (int r = 0; r < 3; r++)
    if (r < head) dst[r] = row[r];
  const uint32_t* src = row + head;
  uint4* dst4 = reinterpret_cast<uint4*>(dst + head);
  const int body = nd - head;
#pragma unroll
  for (int q = 0; q < (SDW + 3) / 4; q++)
    if (4 * q < body) dst4[q] = make_uint4(src[4 * q], src[4 * q + 1], src[4 * q + 2], src[4 * q + 3]);
  if (c0 < 256 && (c1 & 1)) {
    uint16_t* h = reinterpret_cast<uint16_t*>(row);
    h[0] = h[(c0 & 1) + (c1 - c0) - 1];
  }
  return c0 < 256 ? c1 : c0;
}


// ABLATE (tools/microbench/matrix90_ablate only): 1 no drain, 2 keystream only (no parse, no drain),
// 3 parse and drain of a made-up keystream (no AES), 4 parse of a made-up keystream (no AES, no drain)
template <class P, int WG = AES_WG, int ABLATE = 0>
__global__ void __launch_bounds__(WG) k90_sample_matrix(int njobs, const uint8_t* __restrict__ rho, size_t rho_stride,
                                                        int transposed, int16_t* __restrict__ mat) {
  KY_AES_TABLE(tab);
  __shared__ uint32_t stage[WG * STAGE90_DW + 8];  // + 8: drain_private reads whole 16-byte groups
  aes_table_init(tab);
  const AesLds L = aes_lds(tab);
  uint32_t* row = stage + threadIdx.x * STAGE90_DW;
  const uint32_t row_addr = lds_addr(row);
  uint32_t* out32 = reinterpret_cast<uint32_t*>(mat);
  const long job = (long)blockIdx.x * WG + threadIdx.x;
  const bool valid = job < njobs;  // idle lanes carry ctr = 256
  const int jc = valid ? (int)job : njobs - 1;
  const int inst = jc / P::NMAT, e = jc % P::NMAT;
  const int i = e / P::K, j = e % P::K;
  const uint32_t x = transposed ? i : j, y = transposed ? j : i;

  u64p key[4];
  uint32_t rk[60];
  ld32B(rho + (size_t)inst * rho_stride, key);
  aes256_expand(L, key, rk);
  const AesCtr cx = aes256ctr_init(L, rk, x | y << 8);

  int ctr = valid ? 0 : 256;
  // one step: parse 48 bytes of keystream into the lane's staging row, move what it holds to the polynomial
  auto step = [&](const uint32_t (&b0)[4], const uint32_t (&b1)[4], const uint32_t (&b2)[4]) {
    const uint32_t H[4 * XOF90_BLOCKS] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3]};
    if constexpr (ABLATE == 2) {
#pragma unroll
      for (int w = 0; w < 4 * XOF90_BLOCKS; w++) asm volatile("" ::"v"(H[w]));
      ctr += 24;   // eleven steps, the usual number
      return;
    }
    const uint32_t w0 = row_addr + 2u * (ctr & 1);
    uint32_t waddr = w0;
    parse_range<0, XOF90_GROUPS, 4 * XOF90_BLOCKS>(H, waddr);
    const int c0 = ctr, c1 = c0 + (int)((waddr - w0) >> 1);
    if constexpr (ABLATE == 1 || ABLATE == 4) ctr = c1;
    else ctr = drain_private<STAGE90_DW>(row, job, c0, c1, out32);
  };
  if constexpr (AES_NB == 1) {
    for (uint32_t blk = 0; __any(ctr < 256); blk += XOF90_BLOCKS) {
      uint32_t A[XOF90_BLOCKS][4];
#pragma unroll 1
      for (int b = 0; b < XOF90_BLOCKS; b++) {
        uint32_t o[4];
        if constexpr (ABLATE == 3 || ABLATE == 4) {
#pragma unroll
          for (int w = 0; w < 4; w++) o[w] = (rk[8 + w] ^ (blk + b)) * 2654435761u;
        } else {
          aes256ctr_block(L, rk, cx, blk + b, o);
        }
        // A[b] = o, with a compile-time destination (b is a run-time loop index)
#pragma unroll
        for (int bb = 0; bb < XOF90_BLOCKS; bb++)
          if (bb == b) { A[bb][0] = o[0]; A[bb][1] = o[1]; A[bb][2] = o[2]; A[bb][3] = o[3]; }
      }
      step(A[0], A[1], A[2]);
    }
  } else if (__any(ctr < 256)) {
    // -DKY_AES_NB=2: the keystream is produced two blocks at a time (aes256ctr_blocks) and consumed three at a
    // time, so two steps share three pairs; a wavefront that is done after the first of them has computed
    // one block too many.
    for (uint32_t blk = 0;; blk += 6) {
      uint32_t A[2][4], B[2][4];
      aes256ctr_blocks<2>(L, rk, cx, blk, A);
      aes256ctr_blocks<2>(L, rk, cx, blk + 2, B);
      step(A[0], A[1], B[0]);
      if (!__any(ctr < 256)) break;
      aes256ctr_blocks<2>(L, rk, cx, blk + 4, A);
      step(B[1], A[0], A[1]);
      if (!__any(ctr < 256)) break;
    }
  }
}

// nblk keystream blocks from block number `first` on, of n streams: key i, nonce = (nonces[i] as two
// bytes, ten zero bytes), as both callers above build it.    ref/aes256ctr.c:530-586
template <int WG>
__global__ void __launch_bounds__(WG) k_st_aes256ctr(int n, const uint8_t* __restrict__ keys, const uint16_t* __restrict__ nonces,
                                                     uint32_t first, int nblk, uint8_t* __restrict__ out) {
  KY_AES_TABLE(tab);
  aes_table_init(tab);
  const AesLds L = aes_lds(tab);
  const long i = (long)blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  u64p key[4];
  uint32_t rk[60];
  ld32B(keys + (size_t)i * 32, key);
  aes256_expand(L, key, rk);
  const AesCtr cx = aes256ctr_init(L, rk, nonces[i]);
  uint4* o4 = reinterpret_cast<uint4*>(out + (size_t)i * nblk * 16);
  int b = 0;
#pragma unroll 1
  for (; b + 2 <= nblk; b += 2) {
    uint32_t o[2][4];
    aes256ctr_blocks<2>(L, rk, cx, first + (uint32_t)b, o);
    o4[b] = make_uint4(o[0][0], o[0][1], o[0][2], o[0][3]);
    o4[b + 1] = make_uint4(o[1][0], o[1][1], o[1][2], o[1][3]);
  }
  if (b < nblk) {
    uint32_t o[4];
    aes256ctr_block(L, rk, cx, first + (uint32_t)b, o);
    o4[b] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ------------------------------------------------------------------ hash stages (tests)
// the permutation alone on n states of 25 little-endian 64-bit words     ref/fips202.c:82-344
__global__ void __launch_bounds__(SPONGE_WG, 4) k_st_keccak_f1600(int n, uint64_t* __restrict__ st) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64p A[25];
  uint64_t* s = st + (size_t)i * 25;
#pragma unroll
  for (int w = 0; w < 25; w++) A[w] = mk64(s[w]);
  keccak_f1600(A);
#pragma unroll
  for (int w = 0; w < 25; w++) s[w] = to64(A[w]);
}
// H of one pk (CT = false) or one ciphertext (CT = true) per work-item: sha3_256 or, for the 90s
// sets, sha256 (ref/symmetric.h:27,50), through the absorb loops of the FO kernels
template <class P, bool N90, bool CT>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_st_hash_h(int n, const uint8_t* __restrict__ in, uint8_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int LEN = CT ? P::CTB : P::PKB;
  if constexpr (N90) {
    uint32_t h[8];
    u64p hp[4];
    sha256_global<LEN>(in + (size_t)i * LEN, h);
    sha256_digest_le(h, hp);
    st32Bv(out + (size_t)i * 32, hp);
  } else {
    u64p A[25];
    keccak_absorb_words<136, 0x06, LEN>(A, in + (size_t)i * LEN);
    st32B(out + (size_t)i * 32, A[0], A[1], A[2], A[3]);
  }
}

// G of one 32-byte (NW32 = 1: G(d), ref/indcpa.c:216) or 64-byte (NW32 = 2: G(m || H(pk)), ref/kem.c:65,111)
// string per work-item: sha3_512 or, for the 90s sets, sha512 (ref/symmetric.h:28,51), through the device
// functions of the FO kernels
template <int NW32, bool N90>
__global__ void __launch_bounds__(SPONGE_WG, 4) k_st_hash_g(int n, const uint8_t* __restrict__ in, uint8_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64p a[4], b[4], lo[4], hi[4];
  ld32B(in + (size_t)i * 32 * NW32, a);
  if constexpr (NW32 == 2) ld32B(in + (size_t)i * 64 + 32, b);
  if constexpr (N90) {
    sha512_small<NW32>(a, NW32 == 2 ? b : a, lo, hi);
  } else {
    u64p A[25];
    if constexpr (NW32 == 2) {
      sha3_512_64(A, a, b);
    } else {
      keccak_zero(A);
#pragma unroll
      for (int w = 0; w < 4; w++) A[w] = a[w];
      A[4].lo = 0x06;
      A[8].hi = 0x80000000u;
      keccak_f1600(A);
    }
#pragma unroll
    for (int w = 0; w < 4; w++) { lo[w] = A[w]; hi[w] = A[4 + w]; }
  }
  st32Bv(out + (size_t)i * 64, lo);
  st32Bv(out + (size_t)i * 64 + 32, hi);
}

}  // namespace ky
