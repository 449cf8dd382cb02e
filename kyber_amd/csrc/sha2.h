// sha2.h -- SHA-256 / SHA-512 for the -90s parameter sets, one hash state per
// work-item (same "sponge layout" as keccak.h: state in VGPRs, 64 independent
// hashes per wavefront).  Message words are big-endian, so loads are followed
// by a byte swap (v_perm_b32); sigma/Sigma use v_alignbit_b32 rotates and
// v_bitop3_b32 for the three-way xors, Ch and Maj.
//
// Replaces: ref/sha256.c:67-271 and ref/sha512.c:77-302 as used through
// ref/symmetric.h:27-32 (hash_h = sha256, hash_g = sha512, kdf = sha256).
#pragma once
#include "keccak.h"  // bitop3<>, u64p, ld64

namespace ky {

__constant__ const uint32_t SHA256_K[64] = KY_SHA256_K_INIT;
__constant__ const uint32_t SHA256_H0[8] = KY_SHA256_H0_INIT;
__constant__ const uint64_t SHA512_K[80] = KY_SHA512_K_INIT;
__constant__ const uint64_t SHA512_H0[8] = KY_SHA512_H0_INIT;

__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }

// ---------------------------------------------------------------- SHA-256
__device__ __forceinline__ void sha256_init(uint32_t (&h)[8]) {
#pragma unroll
  for (int i = 0; i < 8; i++) h[i] = SHA256_H0[i];
}

// one 64-byte block; w[] holds the 16 big-endian message words and is clobbered
__device__ __forceinline__ void sha256_compress(uint32_t (&h)[8], uint32_t (&w)[16]) {
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll 1
  for (int r0 = 0; r0 < 64; r0 += 16) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (r0) {  // rolling message schedule
        uint32_t x = w[(j + 1) & 15], y = w[(j + 14) & 15];
        uint32_t s0 = bitop3<0x96>(rotr32(x, 7), rotr32(x, 18), x >> 3);
        uint32_t s1 = bitop3<0x96>(rotr32(y, 17), rotr32(y, 19), y >> 10);
        w[j] += s0 + w[(j + 9) & 15] + s1;
      }
      uint32_t S1 = bitop3<0x96>(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
      uint32_t ch = bitop3<0xCA>(e, f, g);
      uint32_t t1 = hh + S1 + ch + SHA256_K[r0 + j] + w[j];
      uint32_t S0 = bitop3<0x96>(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
      uint32_t mj = bitop3<0xE8>(a, b, c);
      hh = g; g = f; f = e; e = d + t1;
      d = c; c = b; b = a; a = t1 + S0 + mj;
    }
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// digest of LEN bytes in global memory (LEN % 32 == 0, 16-byte aligned): the
// lengths hash_h sees are pk (800/1184/1568) and ct (768/1088/1568).
template <int LEN>
__device__ __forceinline__ void sha256_global(const uint8_t* p, uint32_t (&h)[8]) {
  static_assert(LEN % 32 == 0, "");
  uint32_t w[16];
  sha256_init(h);
#pragma unroll 1
  for (int b = 0; b < LEN / 64; b++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      uint4 v = *reinterpret_cast<const uint4*>(p + 64 * b + 16 * q);
      w[4 * q] = bswap32(v.x); w[4 * q + 1] = bswap32(v.y); w[4 * q + 2] = bswap32(v.z); w[4 * q + 3] = bswap32(v.w);
    }
    sha256_compress(h, w);
  }
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = 0;
  if constexpr (LEN % 64 == 32) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      uint4 v = *reinterpret_cast<const uint4*>(p + (LEN / 64) * 64 + 16 * q);
      w[4 * q] = bswap32(v.x); w[4 * q + 1] = bswap32(v.y); w[4 * q + 2] = bswap32(v.z); w[4 * q + 3] = bswap32(v.w);
    }
    w[8] = 0x80000000u;
  } else {
    w[0] = 0x80000000u;
  }
  w[15] = (uint32_t)LEN * 8u;
  sha256_compress(h, w);
}

// 32-byte strings travel between kernels as four little-endian u64p words
__device__ __forceinline__ void le_to_be_words(const u64p (&v)[4], uint32_t* w) {
#pragma unroll
  for (int i = 0; i < 4; i++) { w[2 * i] = bswap32(v[i].lo); w[2 * i + 1] = bswap32(v[i].hi); }
}
__device__ __forceinline__ void sha256_digest_le(const uint32_t (&h)[8], u64p (&out)[4]) {
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = u64p{bswap32(h[2 * i]), bswap32(h[2 * i + 1])};
}

// sha256 of a 32-byte string
__device__ __forceinline__ void sha256_32(const u64p (&in)[4], u64p (&out)[4]) {
  uint32_t h[8], w[16];
  sha256_init(h);
  le_to_be_words(in, w);
  w[8] = 0x80000000u;
#pragma unroll
  for (int i = 9; i < 15; i++) w[i] = 0;
  w[15] = 256;
  sha256_compress(h, w);
  sha256_digest_le(h, out);
}
// sha256 of a 64-byte string a || b                           (kdf, ref/symmetric.h:32)
__device__ __forceinline__ void sha256_64(const u64p (&a)[4], const u64p (&b)[4], u64p (&out)[4]) {
  uint32_t h[8], w[16];
  sha256_init(h);
  le_to_be_words(a, w);
  le_to_be_words(b, w + 8);
  sha256_compress(h, w);
  w[0] = 0x80000000u;
#pragma unroll
  for (int i = 1; i < 15; i++) w[i] = 0;
  w[15] = 512;
  sha256_compress(h, w);
  sha256_digest_le(h, out);
}

// sha256 of a 96-byte string a || b || c                      (kdf over three secrets, ref/kex.c:42,52)
__device__ __forceinline__ void sha256_96(const u64p (&a)[4], const u64p (&b)[4], const u64p (&c)[4], u64p (&out)[4]) {
  uint32_t h[8], w[16];
  sha256_init(h);
  le_to_be_words(a, w);
  le_to_be_words(b, w + 8);
  sha256_compress(h, w);
  le_to_be_words(c, w);
  w[8] = 0x80000000u;
#pragma unroll
  for (int i = 9; i < 15; i++) w[i] = 0;
  w[15] = 768;
  sha256_compress(h, w);
  sha256_digest_le(h, out);
}

// ---------------------------------------------------------------- SHA-512 (single block)
__device__ __forceinline__ uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

// sha512 of NW32 (1 or 2) 32-byte strings -> 64-byte digest as two 32-byte strings
template <int NW32>
__device__ __forceinline__ void sha512_small(const u64p (&a)[4], const u64p (&b)[4], u64p (&lo)[4], u64p (&hi)[4]) {
  uint64_t w[16], s[8];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    w[i] = (uint64_t)bswap32(a[i].lo) << 32 | bswap32(a[i].hi);
    if (NW32 == 2) w[4 + i] = (uint64_t)bswap32(b[i].lo) << 32 | bswap32(b[i].hi);
  }
  w[4 * NW32] = 0x8000000000000000ULL;
  w[15] = 256u * NW32;
#pragma unroll
  for (int i = 0; i < 8; i++) s[i] = SHA512_H0[i];
  uint64_t A = s[0], B = s[1], C = s[2], D = s[3], E = s[4], F = s[5], G = s[6], H = s[7];
#pragma unroll 1
  for (int r0 = 0; r0 < 80; r0 += 16) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (r0) {
        uint64_t x = w[(j + 1) & 15], y = w[(j + 14) & 15];
        w[j] += (rotr64(x, 1) ^ rotr64(x, 8) ^ (x >> 7)) + w[(j + 9) & 15] + (rotr64(y, 19) ^ rotr64(y, 61) ^ (y >> 6));
      }
      uint64_t t1 = H + (rotr64(E, 14) ^ rotr64(E, 18) ^ rotr64(E, 41)) + ((E & F) ^ (~E & G)) + SHA512_K[r0 + j] + w[j];
      uint64_t t2 = (rotr64(A, 28) ^ rotr64(A, 34) ^ rotr64(A, 39)) + ((A & B) ^ (A & C) ^ (B & C));
      H = G; G = F; F = E; E = D + t1;
      D = C; C = B; B = A; A = t1 + t2;
    }
  }
  s[0] += A; s[1] += B; s[2] += C; s[3] += D; s[4] += E; s[5] += F; s[6] += G; s[7] += H;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    lo[i] = u64p{bswap32((uint32_t)(s[i] >> 32)), bswap32((uint32_t)s[i])};
    hi[i] = u64p{bswap32((uint32_t)(s[4 + i] >> 32)), bswap32((uint32_t)s[4 + i])};
  }
}

}  // namespace ky
