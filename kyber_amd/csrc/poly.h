// poly.h -- polynomial layer in "poly layout": 32 work-items own one
// polynomial of Z_3329[X]/(X^256+1), 8 coefficients each.
//
// A work-item t (0..31) of a group holds coefficients in one of three maps:
//   CONTIG : c[k] = coefficient 8t + k            (codecs, basemul, layer len=2)
//   STRIDE : c[m] = coefficient t + 32m           (layers len=128,64,32)
//   MIDDLE : c[m] = coefficient 32(t>>2) + 4m + (t&3)   (layers len=16,8,4)
// so the seven Cooley-Tukey / Gentleman-Sande layers run as three in-register
// passes (3+3+1 layers) with two trips through a 512-byte LDS buffer per
// transform.  Twiddles come from constant memory; the first pass uses
// compile-time indices (scalar loads), the others 2-7 per-lane loads.
//
// Replaces: ref/ntt.c:80-95 ntt, :106-126 invntt, :139-146 basemul,
// ref/poly.c (tobytes/frombytes/compress/decompress/frommsg/tomsg/cbd via
// ref/cbd.c), ref/polyvec.c:15-115 (d=10/11 codecs), :189-201 basemul_acc.
#pragma once
#include "field.h"

namespace ky {

// Rendezvous of the 32 work-items of one polynomial between passes.  They sit
// in one wavefront, whose LDS operations are issued and completed in program
// order, so ordering the compiler is all that is needed.
#define KY_WSYNC()                                            \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
  } while (0)
#define KY_GSYNC() KY_WSYNC()

// Twiddles in LDS for the lane-dependent look-ups (a vector load from the
// constant table costs an L1/L2 round trip per look-up).
__device__ __forceinline__ void twiddles_to_lds(int* Z) {
  if (threadIdx.x < 128) Z[threadIdx.x] = tw_pack(ZETAS[threadIdx.x]);
  __syncthreads();
}
// compile-time twiddle pair of the fixed early layers
#define KY_TW(i) tw_pack(ZETAS_HOST[i])

// ---------------------------------------------------------------- butterflies
// (z is a packed twiddle pair, see fqmul_tw)
__device__ __forceinline__ void ct_bf(int& a, int& b, int z) {  // ref/ntt.c:89-91
  int t = fqmul_tw(b, z);
  b = a - t;
  a = a + t;
}
__device__ __forceinline__ void gs_bf(int& a, int& b, int z) {  // ref/ntt.c:116-119
  int t = a;
  a = barrett(t + b);
  b = fqmul_tw(b - t, z);
}

// ---------------------------------------------------------------- LDS moves
__device__ __forceinline__ void lds_put_contig(int16_t* p, int t, const int (&c)[8]) {
  uint4 v;
  v.x = (uint32_t)(c[0] & 0xFFFF) | (uint32_t)c[1] << 16;
  v.y = (uint32_t)(c[2] & 0xFFFF) | (uint32_t)c[3] << 16;
  v.z = (uint32_t)(c[4] & 0xFFFF) | (uint32_t)c[5] << 16;
  v.w = (uint32_t)(c[6] & 0xFFFF) | (uint32_t)c[7] << 16;
  *reinterpret_cast<uint4*>(p + 8 * t) = v;
}
__device__ __forceinline__ void unpack8(uint4 v, int (&c)[8]) {
  c[0] = (int)(short)v.x; c[1] = (int)v.x >> 16;
  c[2] = (int)(short)v.y; c[3] = (int)v.y >> 16;
  c[4] = (int)(short)v.z; c[5] = (int)v.z >> 16;
  c[6] = (int)(short)v.w; c[7] = (int)v.w >> 16;
}
__device__ __forceinline__ void lds_get_contig(const int16_t* p, int t, int (&c)[8]) {
  unpack8(*reinterpret_cast<const uint4*>(p + 8 * t), c);
}
__device__ __forceinline__ void lds_put_stride(int16_t* p, int t, const int (&c)[8]) {
#pragma unroll
  for (int m = 0; m < 8; m++) p[t + 32 * m] = (int16_t)c[m];
}
__device__ __forceinline__ void lds_get_stride(const int16_t* p, int t, int (&c)[8]) {
#pragma unroll
  for (int m = 0; m < 8; m++) c[m] = p[t + 32 * m];
}
__device__ __forceinline__ void lds_put_middle(int16_t* p, int t, const int (&c)[8]) {
  int base = 32 * (t >> 2) + (t & 3);
#pragma unroll
  for (int m = 0; m < 8; m++) p[base + 4 * m] = (int16_t)c[m];
}
__device__ __forceinline__ void lds_get_middle(const int16_t* p, int t, int (&c)[8]) {
  int base = 32 * (t >> 2) + (t & 3);
#pragma unroll
  for (int m = 0; m < 8; m++) c[m] = p[base + 4 * m];
}

// ---------------------------------------------------------------- forward NTT
// in : polynomial in LDS buffer p (natural order), all 32 work-items arrived
// out: c[] in CONTIG map, Barrett-reduced (= ref/poly.c:244-248 poly_ntt);
//      p is left holding garbage from the intermediate passes.
__device__ __forceinline__ void ntt_fwd(int16_t* p, const int* Z, int t, int (&c)[8]) {
  lds_get_stride(p, t, c);
#pragma unroll
  for (int m = 0; m < 4; m++) ct_bf(c[m], c[m + 4], KY_TW(1));
  ct_bf(c[0], c[2], KY_TW(2)); ct_bf(c[1], c[3], KY_TW(2));
  ct_bf(c[4], c[6], KY_TW(3)); ct_bf(c[5], c[7], KY_TW(3));
  ct_bf(c[0], c[1], KY_TW(4)); ct_bf(c[2], c[3], KY_TW(5));
  ct_bf(c[4], c[5], KY_TW(6)); ct_bf(c[6], c[7], KY_TW(7));
  KY_GSYNC();
  lds_put_stride(p, t, c);
  KY_GSYNC();
  lds_get_middle(p, t, c);
  {
    const int hi = t >> 2;
    int z = Z[8 + hi];
#pragma unroll
    for (int m = 0; m < 4; m++) ct_bf(c[m], c[m + 4], z);
    int z0 = Z[16 + 2 * hi], z1 = Z[17 + 2 * hi];
    ct_bf(c[0], c[2], z0); ct_bf(c[1], c[3], z0);
    ct_bf(c[4], c[6], z1); ct_bf(c[5], c[7], z1);
    ct_bf(c[0], c[1], Z[32 + 4 * hi]); ct_bf(c[2], c[3], Z[33 + 4 * hi]);
    ct_bf(c[4], c[5], Z[34 + 4 * hi]); ct_bf(c[6], c[7], Z[35 + 4 * hi]);
  }
  KY_GSYNC();
  lds_put_middle(p, t, c);
  KY_GSYNC();
  lds_get_contig(p, t, c);
  {
    int z0 = Z[64 + 2 * t], z1 = Z[65 + 2 * t];
    ct_bf(c[0], c[2], z0); ct_bf(c[1], c[3], z0);
    ct_bf(c[4], c[6], z1); ct_bf(c[5], c[7], z1);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = barrett(c[k]);
}

// ---------------------------------------------------------------- inverse NTT
// in : c[] in CONTIG map (NTT domain), p = scratch LDS buffer of the group
// out: c[] in STRIDE map, multiplied by the Montgomery factor
//      (= ref/ntt.c:106-126 invntt, "invntt_tomont")
__device__ __forceinline__ void ntt_inv(int16_t* p, const int* Z, int t, int (&c)[8]) {
  {
    int z0 = Z[127 - 2 * t], z1 = Z[126 - 2 * t];
    gs_bf(c[0], c[2], z0); gs_bf(c[1], c[3], z0);
    gs_bf(c[4], c[6], z1); gs_bf(c[5], c[7], z1);
  }
  KY_GSYNC();
  lds_put_contig(p, t, c);
  KY_GSYNC();
  lds_get_middle(p, t, c);
  {
    const int hi = t >> 2;
    gs_bf(c[0], c[1], Z[63 - 4 * hi]); gs_bf(c[2], c[3], Z[62 - 4 * hi]);
    gs_bf(c[4], c[5], Z[61 - 4 * hi]); gs_bf(c[6], c[7], Z[60 - 4 * hi]);
    int za = Z[31 - 2 * hi], zb = Z[30 - 2 * hi];
    gs_bf(c[0], c[2], za); gs_bf(c[1], c[3], za);
    gs_bf(c[4], c[6], zb); gs_bf(c[5], c[7], zb);
    int z = Z[15 - hi];
#pragma unroll
    for (int m = 0; m < 4; m++) gs_bf(c[m], c[m + 4], z);
  }
  KY_GSYNC();
  lds_put_middle(p, t, c);
  KY_GSYNC();
  lds_get_stride(p, t, c);
  gs_bf(c[0], c[1], KY_TW(7)); gs_bf(c[2], c[3], KY_TW(6));
  gs_bf(c[4], c[5], KY_TW(5)); gs_bf(c[6], c[7], KY_TW(4));
  gs_bf(c[0], c[2], KY_TW(3)); gs_bf(c[1], c[3], KY_TW(3));
  gs_bf(c[4], c[6], KY_TW(2)); gs_bf(c[5], c[7], KY_TW(2));
#pragma unroll
  for (int m = 0; m < 4; m++) gs_bf(c[m], c[m + 4], KY_TW(1));
#pragma unroll
  for (int m = 0; m < 8; m++) c[m] = fqmul_tw(c[m], tw_pack(1441));  // 2^32/128 mod q
}

// STRIDE -> CONTIG through the group's LDS buffer
__device__ __forceinline__ void stride_to_contig(int16_t* p, int t, int (&c)[8]) {
  KY_GSYNC();
  lds_put_stride(p, t, c);
  KY_GSYNC();
  lds_get_contig(p, t, c);
}

// ---------------------------------------------------------------- basemul
// acc[] += a[] o b[] for the work-item's 8 CONTIG coefficients: two products in
// Z_q[X]/(X^2 - zeta) and two in Z_q[X]/(X^2 + zeta).  ref/ntt.c:139-146 as
// driven by ref/poly.c:273-280; summation as ref/polyvec.c:194-198.
__device__ __forceinline__ void basemul_acc(int (&acc)[8], const int (&a)[8], const int (&b)[8], const int* Z, int t) {
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const int z = Z[64 + 2 * t + g];  // twiddle pair
    const int o = 4 * g;
    acc[o + 0] += fqmul_tw(fqmul(a[o + 1], b[o + 1]), z) + fqmul(a[o + 0], b[o + 0]);
    acc[o + 1] += fqmul(a[o + 0], b[o + 1]) + fqmul(a[o + 1], b[o + 0]);
    acc[o + 2] += fqmul_tw(-fqmul(a[o + 3], b[o + 3]), z) + fqmul(a[o + 2], b[o + 2]);  // ref/ntt.c:145: times -zeta
    acc[o + 3] += fqmul(a[o + 2], b[o + 3]) + fqmul(a[o + 3], b[o + 2]);
  }
}

// ---------------------------------------------------------------- noise
// eta = 2: the work-item's 8 coefficients come from one 32-bit word.  ref/cbd.c:58-75
__device__ __forceinline__ void cbd2_word(uint32_t w, int (&c)[8]) {
  uint32_t d = (w & 0x55555555u) + ((w >> 1) & 0x55555555u);
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = (int)((d >> (4 * k)) & 3u) - (int)((d >> (4 * k + 2)) & 3u);
}
// eta = 3: 8 coefficients from 48 bits given as two 24-bit groups.  ref/cbd.c:89-107
__device__ __forceinline__ void cbd3_words(uint32_t lo24, uint32_t hi24, int (&c)[8]) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    uint32_t w = h ? hi24 : lo24;
    uint32_t d = (w & 0x00249249u) + ((w >> 1) & 0x00249249u) + ((w >> 2) & 0x00249249u);
#pragma unroll
    for (int k = 0; k < 4; k++) c[4 * h + k] = (int)((d >> (6 * k)) & 7u) - (int)((d >> (6 * k + 3)) & 7u);
  }
}
// Noise polynomial for work-item t from the PRF output `buf` (ETA*64 bytes, 4-byte aligned)
template <int ETA>
__device__ __forceinline__ void cbd_contig(const uint8_t* buf, int t, int (&c)[8]) {
  if constexpr (ETA == 2) {
    cbd2_word(*reinterpret_cast<const uint32_t*>(buf + 4 * t), c);
  } else {
    const uint16_t* h = reinterpret_cast<const uint16_t*>(buf + 6 * t);
    uint32_t a = h[0], b = h[1], d = h[2];
    cbd3_words(a | (b & 0xFFu) << 16, (b >> 8) | d << 8, c);
  }
}

// ---------------------------------------------------------------- 12-bit codec
// 8 CONTIG coefficients <-> 12 bytes (three dwords).  ref/poly.c:111-145
__device__ __forceinline__ void frombytes12(uint32_t w0, uint32_t w1, uint32_t w2, int (&c)[8]) {
  c[0] = w0 & 0xFFF;
  c[1] = (w0 >> 12) & 0xFFF;
  c[2] = (w0 >> 24 | w1 << 8) & 0xFFF;
  c[3] = (w1 >> 4) & 0xFFF;
  c[4] = (w1 >> 16) & 0xFFF;
  c[5] = (w1 >> 28 | w2 << 4) & 0xFFF;
  c[6] = (w2 >> 8) & 0xFFF;
  c[7] = w2 >> 20;
}
__device__ __forceinline__ void tobytes12(const int (&c)[8], uint32_t& w0, uint32_t& w1, uint32_t& w2) {
  unsigned u[8];
#pragma unroll
  for (int k = 0; k < 8; k++) u[k] = canon(c[k]);
  w0 = u[0] | u[1] << 12 | u[2] << 24;
  w1 = u[2] >> 8 | u[3] << 4 | u[4] << 16 | u[5] << 28;
  w2 = u[5] >> 4 | u[6] << 8 | u[7] << 20;
}

// ---------------------------------------------------------------- d-bit codec
// round(x * 2^D / q) mod 2^D for x in [0,q).  The reference divides
// ((x << D) + q/2) by q (ref/poly.c:30,45; ref/polyvec.c:26,50); here the
// division is a multiply by ceil(2^36/q) = 20642679, exact for every 24-bit
// numerator (checked exhaustively in tests/test_host_logic.py).
__device__ __forceinline__ unsigned div_q(unsigned x) { return (unsigned)(((unsigned long long)x * 20642679ull) >> 36); }

template <int D>
__device__ __forceinline__ unsigned compress1(int a) {
  return div_q((canon(a) << D) + Q / 2) & ((1u << D) - 1);
}
template <int D>
__device__ __forceinline__ int decompress1(unsigned v) {
  return (int)(((v & ((1u << D) - 1)) * Q + (1u << (D - 1))) >> D);
}

// Pack the work-item's 8 coefficients into D bytes at out + D*t (LDS or global).
// Pack the work-item's eight D-bit values into D bytes at out + D*t.
template <int D>
__device__ __forceinline__ void pack_bits_contig(uint8_t* out, int t, const unsigned (&v)[8]) {
  unsigned long long lo = 0, hi = 0;  // 8*D <= 88 bits
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const unsigned long long x = v[k];
    const int pos = D * k;
    if (pos < 64) lo |= x << pos;
    if (pos + D > 64) hi |= (pos >= 64) ? x << (pos - 64) : x >> (64 - pos);
  }
  uint8_t* o = out + D * t;
  // widest stores the alignment of out + D*t allows (`out` is 4-byte aligned):
  // ten single-byte LDS stores per work-item were the kernel's bank-conflict source
  if constexpr (D % 4 == 0) {
#pragma unroll
    for (int b = 0; b < D; b += 4)
      *reinterpret_cast<uint32_t*>(o + b) = (uint32_t)(b < 8 ? lo >> (8 * b) : hi >> (8 * (b - 8)));
  } else if constexpr (D % 2 == 0) {
#pragma unroll
    for (int b = 0; b < D; b += 2)
      *reinterpret_cast<uint16_t*>(o + b) = (uint16_t)(b < 8 ? lo >> (8 * b) : hi >> (8 * (b - 8)));
  } else {
#pragma unroll
    for (int b = 0; b < D; b++) o[b] = (uint8_t)(b < 8 ? lo >> (8 * b) : hi >> (8 * (b - 8)));
  }
}
template <int D>
__device__ __forceinline__ void compress_contig(uint8_t* out, int t, const int (&c)[8]) {
  unsigned v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = compress1<D>(c[k]);
  pack_bits_contig<D>(out, t, v);
}
// Unpack D bytes at in + D*t into the work-item's 8 coefficients.
template <int D>
__device__ __forceinline__ void decompress_contig(const uint8_t* in, int t, int (&c)[8]) {
  const uint8_t* p = in + D * t;
  unsigned long long lo = 0, hi = 0;
  // widest loads the alignment of in + D*t allows (`in` is 4-byte aligned)
  constexpr int W = (D % 4 == 0) ? 4 : (D % 2 == 0) ? 2 : 1;
#pragma unroll
  for (int b = 0; b < D; b += W) {
    unsigned long long x;
    if constexpr (W == 4) x = *reinterpret_cast<const uint32_t*>(p + b);
    else if constexpr (W == 2) x = *reinterpret_cast<const uint16_t*>(p + b);
    else x = p[b];
    if (b < 8) lo |= x << (8 * b);
    else hi |= x << (8 * (b - 8));
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int pos = D * k;
    unsigned long long v = 0;
    if (pos < 64) v = lo >> pos;
    if (pos + D > 64) v |= (pos >= 64) ? hi >> (pos - 64) : hi << (64 - pos);
    c[k] = decompress1<D>((unsigned)v);
  }
}

// The same from an LDS copy of the byte string (`in` 4-byte aligned), with aligned 32-bit loads:
// D bytes starting at any byte offset lie within (D + 6) / 4 consecutive dwords, which are read
// whole and shifted into place.  D = 10 takes three loads instead of five 16-bit ones, D = 11 four
// instead of eleven single bytes.  Reads up to 7 bytes past the field's end: the caller's buffer
// holds the whole ciphertext (the last polynomial, v, has D = 4 or 5, where the excess is 0 or 3
// bytes inside its own 32 D bytes).
template <int D>
__device__ __forceinline__ void decompress_contig_lds32(const uint8_t* in, int t, int (&c)[8]) {
  constexpr int NW = (D % 4 == 0) ? D / 4 : (D + 6) / 4;
  const unsigned o = (unsigned)(D * t);
  const uint32_t* p = reinterpret_cast<const uint32_t*>(in) + (o >> 2);
  const unsigned sh = (o & 3u) * 8u;
  uint32_t w[NW + 1];
#pragma unroll
  for (int i = 0; i < NW; i++) w[i] = p[i];
  w[NW] = 0;
  uint32_t x[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 3 && i < NW; i++) x[i] = (D % 4 == 0) ? w[i] : __builtin_amdgcn_alignbit(w[i + 1], w[i], sh);
  const unsigned long long lo = (unsigned long long)x[1] << 32 | x[0], hi = x[2];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int pos = D * k;
    unsigned long long v = 0;
    if (pos < 64) v = lo >> pos;
    if (pos + D > 64) v |= (pos >= 64) ? hi >> (pos - 64) : hi << (64 - pos);
    c[k] = decompress1<D>((unsigned)v);
  }
}

// ---------------------------------------------------------------- message
// byte t of the message <-> coefficients 8t..8t+7.  ref/poly.c:155-194
__device__ __forceinline__ void frommsg_contig(unsigned byte, int (&c)[8]) {
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = ((byte >> k) & 1u) ? (Q + 1) / 2 : 0;
}
__device__ __forceinline__ unsigned tomsg_contig(const int (&c)[8]) {
  unsigned b = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) b |= (div_q((canon(c[k]) << 1) + Q / 2) & 1u) << k;
  return b;
}

}  // namespace ky
