// aes.h -- AES-256 in counter mode for the -90s parameter sets, one cipher
// context per work-item.
//
// The reference runs a bitsliced constant-time AES four blocks at a time
// (ref/aes256ctr.c:339-520).  On gfx950 the natural form is the classic
// 32-bit "T-table" round: one 1 KiB table T0[x] = (2*S[x], S[x], S[x], 3*S[x])
// kept in LDS, the other three tables being byte rotations of it
// (v_alignbit_b32).  The table is stored 16 times, interleaved so that lane l
// reads copy l % 16: the two lanes of a 32-lane access group that share a copy
// can collide at most two ways, whatever the (data-dependent) indices are.
// Table addresses depend on key/state bytes, i.e. this is not a constant-time
// AES on hardware whose LDS timing depends on bank conflicts; the matrix XOF
// runs on public data, the PRF does not (recorded in DESIGN.md).
//
// Counter layout restated from ref/aes256ctr.c:530-543,565-577: block input =
// 12-byte nonce || 32-bit big-endian block counter.
#pragma once
#include "keccak.h"
#include "sha2.h"

namespace ky {

__constant__ const uint8_t AES_SBOX[256] = KY_AES_SBOX_INIT;

constexpr int AES_TCOPIES = 16;
constexpr int AES_TABLE_DW = 256 * AES_TCOPIES;  // 16 KiB of LDS

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

struct AesLds {
  const uint32_t* tab;  // this lane's copy: tab[x * AES_TCOPIES]
  __device__ __forceinline__ uint32_t t0(uint32_t x) const { return tab[x * AES_TCOPIES]; }
  __device__ __forceinline__ uint32_t sbox(uint32_t x) const { return (t0(x) >> 8) & 0xFFu; }
};
__device__ __forceinline__ AesLds aes_lds(const uint32_t* tab) { return AesLds{tab + (threadIdx.x & (AES_TCOPIES - 1))}; }

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }

__device__ __forceinline__ uint32_t aes_subword(const AesLds& L, uint32_t w) {
  return L.sbox(w & 0xFF) | L.sbox((w >> 8) & 0xFF) << 8 | L.sbox((w >> 16) & 0xFF) << 16 | L.sbox(w >> 24) << 24;
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
    uint32_t t0 = L.t0(s0 & 0xFF) ^ rotl32(L.t0((s1 >> 8) & 0xFF), 8) ^ rotl32(L.t0((s2 >> 16) & 0xFF), 16) ^ rotl32(L.t0(s3 >> 24), 24) ^ rk[4 * r];
    uint32_t t1 = L.t0(s1 & 0xFF) ^ rotl32(L.t0((s2 >> 8) & 0xFF), 8) ^ rotl32(L.t0((s3 >> 16) & 0xFF), 16) ^ rotl32(L.t0(s0 >> 24), 24) ^ rk[4 * r + 1];
    uint32_t t2 = L.t0(s2 & 0xFF) ^ rotl32(L.t0((s3 >> 8) & 0xFF), 8) ^ rotl32(L.t0((s0 >> 16) & 0xFF), 16) ^ rotl32(L.t0(s1 >> 24), 24) ^ rk[4 * r + 2];
    uint32_t t3 = L.t0(s3 & 0xFF) ^ rotl32(L.t0((s0 >> 8) & 0xFF), 8) ^ rotl32(L.t0((s1 >> 16) & 0xFF), 16) ^ rotl32(L.t0(s2 >> 24), 24) ^ rk[4 * r + 3];
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
  out[0] = (L.sbox(s0 & 0xFF) | L.sbox((s1 >> 8) & 0xFF) << 8 | L.sbox((s2 >> 16) & 0xFF) << 16 | L.sbox(s3 >> 24) << 24) ^ rk[56];
  out[1] = (L.sbox(s1 & 0xFF) | L.sbox((s2 >> 8) & 0xFF) << 8 | L.sbox((s3 >> 16) & 0xFF) << 16 | L.sbox(s0 >> 24) << 24) ^ rk[57];
  out[2] = (L.sbox(s2 & 0xFF) | L.sbox((s3 >> 8) & 0xFF) << 8 | L.sbox((s0 >> 16) & 0xFF) << 16 | L.sbox(s1 >> 24) << 24) ^ rk[58];
  out[3] = (L.sbox(s3 & 0xFF) | L.sbox((s0 >> 8) & 0xFF) << 8 | L.sbox((s1 >> 16) & 0xFF) << 16 | L.sbox(s2 >> 24) << 24) ^ rk[59];
}

// keystream block number ctr for nonce word n0 (nonce bytes 4..11 are zero in
// every use the KEM makes: ref/symmetric-aes.c:7-20)
__device__ __forceinline__ void aes256ctr_block(const AesLds& L, const uint32_t (&rk)[60], uint32_t n0, uint32_t ctr, uint32_t (&out)[4]) {
  const uint32_t in[4] = {n0, 0u, 0u, bswap32(ctr)};
  aes256_encrypt(L, rk, in, out);
}

}  // namespace ky
