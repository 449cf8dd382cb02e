// aes.h -- AES-256 in counter mode for the -90s parameter sets, one cipher
// context per work-item.
//
// The reference runs a bitsliced constant-time AES four blocks at a time
// (ref/aes256ctr.c:339-520).  On gfx950 the natural form is the classic
// 32-bit "T-table" round: one 1 KiB table T0[x] = (2*S[x], S[x], S[x], 3*S[x])
// kept in LDS, the other three tables being byte rotations of it
// (v_alignbit_b32).  The table is stored 32 times, interleaved (entry x of copy
// c at dword 32 x + c), and lane l reads copy l % 32.  ds_read_b32 serves a
// wavefront as two groups of 32 lanes with bank = dword address mod 32
// (MI355X_MICROARCH.md, LDS table), so every lane of a group owns one bank
// whatever its (key- and data-dependent) index is: the look-ups are free of bank
// conflicts, which also makes their timing independent of the secret PRF keys.
//
// Counter layout restated from ref/aes256ctr.c:530-543,565-577: block input =
// 12-byte nonce || 32-bit big-endian block counter.
#pragma once
#include "keccak.h"
#include "sha2.h"

namespace ky {

__constant__ const uint8_t AES_SBOX[256] = KY_AES_SBOX_INIT;

constexpr int AES_TCOPIES = 32;  // ds_read_b32 banks are (a/4) mod 32 per 32-lane group: one private bank per lane
constexpr int AES_TSHIFT = 7;    // log2(4 * AES_TCOPIES): byte offset of entry x is x << AES_TSHIFT
constexpr int AES_TABLE_DW = 256 * AES_TCOPIES;  // 32 KiB of LDS

__device__ __forceinline__ uint32_t aes_xtime(uint32_t s) { return ((s << 1) ^ ((s & 0x80u) ? 0x1Bu : 0u)) & 0xFFu; }

// every work-item of the workgroup takes part; ends with a barrier
__device__ __forceinline__ void aes_table_init(uint32_t* tab) {
  for (int x = threadIdx.x; x < 256; x += blockDim.x) {
    uint32_t s = AES_SBOX[x], s2 = aes_xtime(s), s3 = s2 ^ s;
    uint32_t t = s2 | s << 8 | s << 16 | s3 << 24;
#pragma unroll
    for (int c = 0; c < AES_TCOPIES; c++) tab[x * AES_TCOPIES + c] = t;
  }
  __syncthreads();
}

// The table is aligned to its own size (32 KiB) in LDS, so the byte address of
// lane l's copy of entry x is  base | x << 7  with base = table + 4 * (l % 32):
// one shift and one v_and_or_b32 per look-up instead of extract, scale, add.
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
struct AesLds {
  uint32_t base;  // LDS byte address of this lane's copy of entry 0
  __device__ __forceinline__ uint32_t at(uint32_t off) const { return *(lds_cu32*)(uintptr_t)(off | base); }
  // T0[byte K of s]
  template <int K>
  __device__ __forceinline__ uint32_t look(uint32_t s) const {
    constexpr uint32_t M = 0xFFu << AES_TSHIFT;
    if constexpr (K == 0) return at((s << AES_TSHIFT) & M);
    else return at((s >> (8 * K - AES_TSHIFT)) & M);
  }
  __device__ __forceinline__ uint32_t t0(uint32_t x) const { return at(x << AES_TSHIFT); }
  __device__ __forceinline__ uint32_t sbox(uint32_t x) const { return (t0(x) >> 8) & 0xFFu; }
};
#define KY_AES_TABLE(name) __shared__ __attribute__((aligned(4 * AES_TABLE_DW))) uint32_t name[AES_TABLE_DW]
__device__ __forceinline__ AesLds aes_lds(const uint32_t* tab) {
  return AesLds{(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)tab + 4u * (threadIdx.x & (AES_TCOPIES - 1))};
}

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }

__device__ __forceinline__ uint32_t aes_subword(const AesLds& L, uint32_t w) {
  return L.sbox(w & 0xFF) | L.sbox((w >> 8) & 0xFF) << 8 | L.sbox((w >> 16) & 0xFF) << 16 | L.sbox(w >> 24) << 24;
}

// one output column of a middle round: T0[a.b0] ^ rotl8 T0[b.b1] ^ rotl16 T0[c.b2] ^ rotl24 T0[d.b3] ^ k
__device__ __forceinline__ uint32_t aes_col(const AesLds& L, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t k) {
  const uint32_t x0 = L.look<0>(a), x1 = rotl32(L.look<1>(b), 8), x2 = rotl32(L.look<2>(c), 16), x3 = rotl32(L.look<3>(d), 24);
  return bitop3<0x96>(bitop3<0x96>(x0, x1, x2), x3, k);
}
// one output column of the last round: S[a.b0] | S[b.b1] << 8 | S[c.b2] << 16 | S[d.b3] << 24, then ^ k.
// S sits in bytes 1 and 2 of a table entry, so three byte permutes assemble the word.
__device__ __forceinline__ uint32_t aes_col_last(const AesLds& L, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t k) {
  const uint32_t e0 = L.look<0>(a), e1 = L.look<1>(b), e2 = L.look<2>(c), e3 = L.look<3>(d);
  const uint32_t p01 = __builtin_amdgcn_perm(e1, e0, 0x0c0c0501u);  // [e0.b1, e1.b1, 0, 0]
  const uint32_t p23 = __builtin_amdgcn_perm(e3, e2, 0x06020c0cu);  // [0, 0, e2.b2, e3.b2]
  return (p01 | p23) ^ k;
}

// FIPS 197 §5.2 on little-endian packed words (byte 0 of a word = row 0)
__device__ __forceinline__ void aes256_expand(const AesLds& L, const u64p (&key)[4], uint32_t (&rk)[60]) {
#pragma unroll
  for (int i = 0; i < 4; i++) { rk[2 * i] = key[i].lo; rk[2 * i + 1] = key[i].hi; }
  uint32_t rcon = 1;
#pragma unroll
  for (int i = 8; i < 60; i++) {
    uint32_t t = rk[i - 1];
    if (i % 8 == 0) {
      t = aes_subword(L, rotr32(t, 8)) ^ rcon;
      rcon = aes_xtime(rcon);
    } else if (i % 8 == 4) {
      t = aes_subword(L, t);
    }
    rk[i] = rk[i - 8] ^ t;
  }
}

// one block: s = AES-256_rk(in); state words are columns, little-endian packed
__device__ __forceinline__ void aes256_encrypt(const AesLds& L, const uint32_t (&rk)[60], const uint32_t (&in)[4], uint32_t (&out)[4]) {
  uint32_t s0 = in[0] ^ rk[0], s1 = in[1] ^ rk[1], s2 = in[2] ^ rk[2], s3 = in[3] ^ rk[3];
#pragma unroll
  for (int r = 1; r < 14; r++) {
    // column c <- T0[row0 of col c] ^ T1[row1 of col c+1] ^ T2[row2 of col c+2] ^ T3[row3 of col c+3] ^ rk
    const uint32_t t0 = aes_col(L, s0, s1, s2, s3, rk[4 * r]);
    const uint32_t t1 = aes_col(L, s1, s2, s3, s0, rk[4 * r + 1]);
    const uint32_t t2 = aes_col(L, s2, s3, s0, s1, rk[4 * r + 2]);
    const uint32_t t3 = aes_col(L, s3, s0, s1, s2, rk[4 * r + 3]);
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
  out[0] = aes_col_last(L, s0, s1, s2, s3, rk[56]);
  out[1] = aes_col_last(L, s1, s2, s3, s0, rk[57]);
  out[2] = aes_col_last(L, s2, s3, s0, s1, rk[58]);
  out[3] = aes_col_last(L, s3, s0, s1, s2, rk[59]);
}

// Counter mode with the KEM's nonces (bytes 4..11 zero: ref/symmetric-aes.c:7-20):
// the input block is (n0, 0, 0, bswap(ctr)), so after the first AddRoundKey three
// of the four state words are the same for every block of a stream.  Round 1
// has exactly one look-up per output column that depends on the counter word;
// the other twelve are folded once per stream into p[] (with the round key).
struct AesCtr {
  uint32_t p[4];
};
__device__ __forceinline__ AesCtr aes256ctr_init(const AesLds& L, const uint32_t (&rk)[60], uint32_t n0) {
  const uint32_t s0 = n0 ^ rk[0], s1 = rk[1], s2 = rk[2];
  AesCtr c;
  c.p[0] = bitop3<0x96>(L.look<0>(s0), rotl32(L.look<1>(s1), 8), rotl32(L.look<2>(s2), 16)) ^ rk[4];   // + rotl24 T0[s3.b3]
  c.p[1] = bitop3<0x96>(L.look<0>(s1), rotl32(L.look<1>(s2), 8), rotl32(L.look<3>(s0), 24)) ^ rk[5];   // + rotl16 T0[s3.b2]
  c.p[2] = bitop3<0x96>(L.look<0>(s2), rotl32(L.look<2>(s0), 16), rotl32(L.look<3>(s1), 24)) ^ rk[6];  // + rotl8  T0[s3.b1]
  c.p[3] = bitop3<0x96>(rotl32(L.look<1>(s0), 8), rotl32(L.look<2>(s1), 16), rotl32(L.look<3>(s2), 24)) ^ rk[7];  // + T0[s3.b0]
  return c;
}
// keystream block number ctr
__device__ __forceinline__ void aes256ctr_block(const AesLds& L, const uint32_t (&rk)[60], const AesCtr& c, uint32_t ctr, uint32_t (&out)[4]) {
  const uint32_t v = bswap32(ctr) ^ rk[3];
  uint32_t s0 = c.p[0] ^ rotl32(L.look<3>(v), 24), s1 = c.p[1] ^ rotl32(L.look<2>(v), 16), s2 = c.p[2] ^ rotl32(L.look<1>(v), 8),
           s3 = c.p[3] ^ L.look<0>(v);
#pragma unroll
  for (int r = 2; r < 14; r++) {
    const uint32_t t0 = aes_col(L, s0, s1, s2, s3, rk[4 * r]);
    const uint32_t t1 = aes_col(L, s1, s2, s3, s0, rk[4 * r + 1]);
    const uint32_t t2 = aes_col(L, s2, s3, s0, s1, rk[4 * r + 2]);
    const uint32_t t3 = aes_col(L, s3, s0, s1, s2, rk[4 * r + 3]);
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
  out[0] = aes_col_last(L, s0, s1, s2, s3, rk[56]);
  out[1] = aes_col_last(L, s1, s2, s3, s0, rk[57]);
  out[2] = aes_col_last(L, s2, s3, s0, s1, rk[58]);
  out[3] = aes_col_last(L, s3, s0, s1, s2, rk[59]);
}

}  // namespace ky
