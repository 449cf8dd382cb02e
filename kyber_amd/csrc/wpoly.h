// wpoly.h -- "wide" poly layout for the standalone polynomial kernels (BASELINE.json config 2):
// 16 work-items own one polynomial of Z_3329[X]/(X^256+1), 16 coefficients each, four polynomials
// per wavefront.  Same exact-integer fp32 arithmetic as fpoly.h; what changes is the shape:
//
//   * Four index bits are local to a work-item, so the seven butterfly layers are 1 + 4 + 2 in
//     three in-register passes with TWO transposes through LDS per transform (fpoly.h: three).
//       SPLIT (S): c[8h + k] = coefficient 128h + 8l + k      local bits 7,2,1,0   lane l = bits 6..3
//       MID   (M): c[m]      = coefficient 128h + 8m + low3    local bits 6..3      lane   = (h, low3)
//     S is also the memory layout: a work-item moves two contiguous 16-byte pieces, the first and
//     the second half of the polynomial, so every global access is a full 256-byte run per 16 lanes.
//   * Every lane-dependent twiddle (21 forward, 21 inverse) sits in registers as a (zeta, zeta/q)
//     pair: nothing is looked up inside a transform, no workgroup-shared table, no workgroup barrier.
//   * A product by a constant is three FMAs (fmulc), a butterfly five full-rate instructions.
//   * Outputs are canonical ([0, q)): the last reduction takes the floor quotient (fred_canon, one
//     instruction more than the centred form) instead of a fix-up after packing.
//
// Replaces (behaviour): ref/ntt.c:80-95 ntt, :106-126 invntt, :139-146 basemul as driven by
// ref/poly.c:244-280, each followed by the canonicalisation every serializer of the reference
// applies (ref/poly.c:118-121).
#pragma once
#include "fpoly.h"

namespace ky {

// ---------------------------------------------------------------- products by constants
struct Tw {
  float z, zq;  // twiddle (centred residue) and twiddle / q
};
__device__ __forceinline__ Tw mk_tw(float z) { return Tw{z, z * QINVF}; }

constexpr float QM = 3329.0f * 12582912.0f;  // q * RND_MAGIC = 9987 * 2^22, an fp32 number

// z * b mod q for |z * b| < 2^24: k = RND_MAGIC + rint(b z / q), m = -q rint(b z / q) (exact: the
// fma's exact result is an integer below 2^24), r = z b + m.  The quotient may be off by one next to
// a half (z/q is rounded), so |r| <= 1665.
__device__ __forceinline__ float fmulc(float b, Tw w) {
  const float k = __builtin_fmaf(b, w.zq, RND_MAGIC);
  const float m = __builtin_fmaf(k, -QF, QM);
  return __builtin_fmaf(w.z, b, m);
}
// The same two with the result in [0, q): the quotient of x - (q-1)/2, rounded to nearest, is
// floor(x / q), because (x - 1664) / q is never within 1 / 2q of a half while the rounded 1/q puts
// it off by |x| * 2.44e-8 < 0.41 / q for |x| < 2^24.  One instruction more than the centred forms and
// nothing to fix up when the coefficient is packed.
__device__ __forceinline__ float fred_canon(float x) {
  const float k = __builtin_fmaf(x - 1664.0f, QINVF, RND_MAGIC);
  return x + __builtin_fmaf(k, -QF, QM);
}
__device__ __forceinline__ float fmulc_canon(float b, float z) {
  const float k = __builtin_fmaf(__builtin_fmaf(z, b, -1664.0f), QINVF, RND_MAGIC);
  return __builtin_fmaf(z, b, __builtin_fmaf(k, -QF, QM));
}
__device__ __forceinline__ void w_ct(float& a, float& b, Tw w) {  // ref/ntt.c:89-91
  const float t = fmulc(b, w);
  b = a - t;
  a = a + t;
}
__device__ __forceinline__ void w_gs(float& a, float& b, Tw w) {  // ref/ntt.c:116-119, sum left unreduced
  const float d = b - a;
  a = a + b;
  b = fmulc(d, w);
}

// ---------------------------------------------------------------- lane context: LDS slots
// Coefficient i of the wavefront's polynomial p (0..3) lives at dword
//     256 p + (i ^ (bit7(i) << 3) ^ (par << 4) ^ (par << 2)),   par = p & 1
// which makes the 16-byte accesses of the S side hit 64 distinct banks per 16-lane group and the
// 4-byte accesses of the M side 32 distinct banks per 32-lane group (the 16-byte stores are two-way,
// which costs 3 of their 13 cycles).
struct WLane {
  int l;         // work-item within the polynomial, 0..15
  float* sq[4];  // S side: the quad of (h, j) = coefficients 128h + 8l + 4j .. +3 at sq[2h + j]
  float* mq[4];  // M side: coefficient 128h + 8m + low3 at mq[m & 3] + 32 (m >> 2), (h, low3) = (l >> 3, l & 7)
};
__device__ __forceinline__ WLane w_lane(float* wave_buf) {
  const int lane = (int)threadIdx.x & 63, p = lane >> 4, par = p & 1;
  WLane L;
  L.l = lane & 15;
  float* base = wave_buf + 256 * p;
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int j = 0; j < 2; j++) L.sq[2 * h + j] = base + 128 * h + 8 * (L.l ^ h ^ (2 * par)) + 4 * (j ^ par);
  const int h = L.l >> 3, low3 = L.l & 7;
#pragma unroll
  for (int ml = 0; ml < 4; ml++) L.mq[ml] = base + 128 * h + 8 * (ml ^ h ^ (2 * par)) + (low3 ^ (4 * par));
  return L;
}
__device__ __forceinline__ void w_put_s(const WLane& L, const float (&c)[16]) {
#pragma unroll
  for (int q = 0; q < 4; q++) *reinterpret_cast<float4*>(L.sq[q]) = make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
}
__device__ __forceinline__ void w_get_s(const WLane& L, float (&c)[16]) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float4 v = *reinterpret_cast<const float4*>(L.sq[q]);
    c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
  }
}
__device__ __forceinline__ void w_put_m(const WLane& L, const float (&c)[16]) {
#pragma unroll
  for (int m = 0; m < 16; m++) L.mq[m & 3][32 * (m >> 2)] = c[m];
}
__device__ __forceinline__ void w_get_m(const WLane& L, float (&c)[16]) {
#pragma unroll
  for (int m = 0; m < 16; m++) c[m] = L.mq[m & 3][32 * (m >> 2)];
}
// the 64 lanes of a wavefront exchange through its own 4 KB: ordering the compiler is enough
#define KY_W_S2M(L, c) do { KY_WSYNC(); w_put_s(L, c); KY_WSYNC(); w_get_m(L, c); } while (0)
#define KY_W_M2S(L, c) do { KY_WSYNC(); w_put_m(L, c); KY_WSYNC(); w_get_s(L, c); } while (0)

__device__ __forceinline__ void w_fred16(float (&c)[16]) {
#pragma unroll
  for (int k = 0; k < 16; k++) c[k] = fred(c[k]);
}
// Inputs beyond a kernel's entry bound are centred first.  Rare, and decided per wavefront: the empty
// asm keeps the compiler from turning the branch into 16 reductions plus 16 selects on every call.
__device__ __forceinline__ void w_centre_if(bool any_wide, float (&c)[16]) {
  if (any_wide) {
    asm volatile("");
    w_fred16(c);
  }
}
__device__ __forceinline__ void w_fred16_canon(float (&c)[16]) {
#pragma unroll
  for (int k = 0; k < 16; k++) c[k] = fred_canon(c[k]);
}

// ---------------------------------------------------------------- twiddles of a lane
// forward, index = 2^layer + (coefficient index >> (8 - layer)), ref/ntt.c:84-86
struct WTwFwd {
  Tw z64, z32[2], z16[4], z8[8];  // M pass (by h = l >> 3 and the local m)
  Tw z4[2], z2[4];                // S pass (by l; z4[h], z2[2h + j]); z2 are the base-multiplication twiddles too
};
__device__ __forceinline__ void w_tw_fwd(WTwFwd& T, int l) {
  const int h = l >> 3;
  T.z64 = mk_tw(ZETAF[2 + h]);
  const float2 a = *reinterpret_cast<const float2*>(&ZETAF[4 + 2 * h]);
  T.z32[0] = mk_tw(a.x); T.z32[1] = mk_tw(a.y);
  const float4 b = *reinterpret_cast<const float4*>(&ZETAF[8 + 4 * h]);
  T.z16[0] = mk_tw(b.x); T.z16[1] = mk_tw(b.y); T.z16[2] = mk_tw(b.z); T.z16[3] = mk_tw(b.w);
  const float4 c0 = *reinterpret_cast<const float4*>(&ZETAF[16 + 8 * h]), c1 = *reinterpret_cast<const float4*>(&ZETAF[20 + 8 * h]);
  T.z8[0] = mk_tw(c0.x); T.z8[1] = mk_tw(c0.y); T.z8[2] = mk_tw(c0.z); T.z8[3] = mk_tw(c0.w);
  T.z8[4] = mk_tw(c1.x); T.z8[5] = mk_tw(c1.y); T.z8[6] = mk_tw(c1.z); T.z8[7] = mk_tw(c1.w);
  T.z4[0] = mk_tw(ZETAF[32 + l]); T.z4[1] = mk_tw(ZETAF[48 + l]);
  const float2 d0 = *reinterpret_cast<const float2*>(&ZETAF[64 + 2 * l]), d1 = *reinterpret_cast<const float2*>(&ZETAF[96 + 2 * l]);
  T.z2[0] = mk_tw(d0.x); T.z2[1] = mk_tw(d0.y); T.z2[2] = mk_tw(d1.x); T.z2[3] = mk_tw(d1.y);
}
// inverse, index = 2^(layer+1) - 1 - (coefficient index >> (8 - layer)), ref/ntt.c:110-113
struct WTwInv {
  Tw z2[4], z4[2];                // S pass
  Tw z8[8], z16[4], z32[2], z64;  // M pass
};
__device__ __forceinline__ void w_tw_inv(WTwInv& T, int l) {
  const int h = l >> 3;
  const float2 d0 = *reinterpret_cast<const float2*>(&ZETAF[126 - 2 * l]), d1 = *reinterpret_cast<const float2*>(&ZETAF[94 - 2 * l]);
  T.z2[0] = mk_tw(d0.y); T.z2[1] = mk_tw(d0.x); T.z2[2] = mk_tw(d1.y); T.z2[3] = mk_tw(d1.x);
  T.z4[0] = mk_tw(ZETAF[63 - l]); T.z4[1] = mk_tw(ZETAF[47 - l]);
  const float4 c0 = *reinterpret_cast<const float4*>(&ZETAF[24 - 8 * h]), c1 = *reinterpret_cast<const float4*>(&ZETAF[28 - 8 * h]);
  T.z8[0] = mk_tw(c1.w); T.z8[1] = mk_tw(c1.z); T.z8[2] = mk_tw(c1.y); T.z8[3] = mk_tw(c1.x);
  T.z8[4] = mk_tw(c0.w); T.z8[5] = mk_tw(c0.z); T.z8[6] = mk_tw(c0.y); T.z8[7] = mk_tw(c0.x);
  const float4 b = *reinterpret_cast<const float4*>(&ZETAF[12 - 4 * h]);
  T.z16[0] = mk_tw(b.w); T.z16[1] = mk_tw(b.z); T.z16[2] = mk_tw(b.y); T.z16[3] = mk_tw(b.x);
  const float2 a = *reinterpret_cast<const float2*>(&ZETAF[6 - 2 * h]);
  T.z32[0] = mk_tw(a.y); T.z32[1] = mk_tw(a.x);
  T.z64 = mk_tw(ZETAF[3 - h]);
}

// ---------------------------------------------------------------- forward transform
// in : c[] in S, |c| <= 4095           out: c[] in S, plain NTT (bit-reversed order as ref/ntt.c), NOT reduced:
// |c| <= 8325, every value that entered a product was <= 9090 (15.1 M < 2^24).
// bounds: a layer adds <= 1665: 4095 -> 5760 (S) -> 7425 -> 9090, reduce -> 1665 -> 3330 -> 4995 (M)
// -> 6660 -> 8325 (S).
__device__ __forceinline__ void w_ntt_fwd(const WLane& L, const WTwFwd& T, float (&c)[16]) {
  const Tw z128 = mk_tw(ZETAF[1]);
#pragma unroll
  for (int k = 0; k < 8; k++) w_ct(c[k], c[k + 8], z128);
  KY_W_S2M(L, c);
#pragma unroll
  for (int m = 0; m < 8; m++) w_ct(c[m], c[m + 8], T.z64);
#pragma unroll
  for (int g = 0; g < 2; g++)
#pragma unroll
    for (int m = 0; m < 4; m++) w_ct(c[8 * g + m], c[8 * g + m + 4], T.z32[g]);
  w_fred16(c);
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int m = 0; m < 2; m++) w_ct(c[4 * g + m], c[4 * g + m + 2], T.z16[g]);
#pragma unroll
  for (int g = 0; g < 8; g++) w_ct(c[2 * g], c[2 * g + 1], T.z8[g]);
  KY_W_M2S(L, c);
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int u = 0; u < 4; u++) w_ct(c[8 * h + u], c[8 * h + 4 + u], T.z4[h]);
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int u = 0; u < 2; u++) w_ct(c[4 * g + u], c[4 * g + 2 + u], T.z2[g]);
}

// ---------------------------------------------------------------- inverse transform
// in : c[] in S (NTT domain), |c| <= 1665
// out: c[] in S, scale * INTT(c) as canonical residues in [0, q), where the plain inverse transform
//      carries 128^-1: scale = 1 for a true product, 2^16 for the reference's invntt_tomont
//      (ref/ntt.c:106-126: 1441 = 2^32 / 128).  fin = scale / 128, gin = fin * zeta[1] (centred).
// bounds: a sum doubles; a difference that enters a product is kept <= 6660 (11.1 M) by reducing
// the sums that have doubled twice.
__device__ __forceinline__ void w_ntt_inv(const WLane& L, const WTwInv& T, float (&c)[16], float fin, float gin) {
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int u = 0; u < 2; u++) w_gs(c[4 * g + u], c[4 * g + 2 + u], T.z2[g]);  // sums 3330
#pragma unroll
  for (int h = 0; h < 2; h++) {
#pragma unroll
    for (int u = 0; u < 4; u++) w_gs(c[8 * h + u], c[8 * h + 4 + u], T.z4[h]);  // u < 2: 6660, u >= 2: 3330
    c[8 * h] = fred(c[8 * h]);
    c[8 * h + 1] = fred(c[8 * h + 1]);
  }
  KY_W_S2M(L, c);  // every value <= 3330
#pragma unroll
  for (int g = 0; g < 8; g++) {
    w_gs(c[2 * g], c[2 * g + 1], T.z8[g]);  // sums 6660
    c[2 * g] = fred(c[2 * g]);
  }
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int m = 0; m < 2; m++) w_gs(c[4 * g + m], c[4 * g + m + 2], T.z16[g]);  // sums 3330
#pragma unroll
  for (int g = 0; g < 2; g++) {
#pragma unroll
    for (int m = 0; m < 4; m++) w_gs(c[8 * g + m], c[8 * g + m + 4], T.z32[g]);  // m < 2: 6660, else 3330
    c[8 * g] = fred(c[8 * g]);
    c[8 * g + 1] = fred(c[8 * g + 1]);
  }
#pragma unroll
  for (int m = 0; m < 8; m++) w_gs(c[m], c[m + 8], T.z64);  // m = 2, 3: 6660, others <= 3330
  c[2] = fred(c[2]);
  c[3] = fred(c[3]);
  KY_W_M2S(L, c);  // every value <= 3330
#pragma unroll
  for (int k = 0; k < 8; k++) {  // last layer with the scaling folded in: both outputs are products
    const float s = c[k] + c[k + 8], d = c[k + 8] - c[k];
    c[k] = fmulc_canon(s, fin);  // |s|, |d| <= 6660: products <= 11.1 M
    c[k + 8] = fmulc_canon(d, gin);
  }
}

// ---------------------------------------------------------------- base multiplication
// r = a o b in S: per quad (c0 + c1 X)(d0 + d1 X) mod X^2 - zeta and mod X^2 + zeta, ref/ntt.c:139-146
// as driven by ref/poly.c:273-280.  |a| <= 1665; |b| <= 8325 (an unreduced w_ntt_fwd output).
// out: |r| < 2^24, not reduced.  Bounds: zeta b1 <= 13.9 M; a0 b0 + a1 (zeta b1 mod q) <= 13.86 M + 2.77 M;
// a0 b1 is reduced before a1 b0 joins it.
__device__ __forceinline__ void w_basemul(float (&r)[16], const float (&a)[16], const float (&b)[16], const WTwFwd& T) {
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const Tw zp = T.z2[g], zn = Tw{-zp.z, -zp.zq};
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int o = 4 * g + 2 * s;
      const float zb = fmulc(b[o + 1], s ? zn : zp);
      r[o] = __builtin_fmaf(a[o + 1], zb, a[o] * b[o]);
      r[o + 1] = __builtin_fmaf(a[o + 1], b[o], fred(a[o] * b[o + 1]));
    }
  }
}

// ---------------------------------------------------------------- memory side
typedef uint32_t w_u32x4 __attribute__((ext_vector_type(4)));
// streamed once: keep the lines out of the way of nothing (non-temporal), 16 bytes per lane
__device__ __forceinline__ uint4 w_ld16(const int16_t* p) {
  const w_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const w_u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void w_st16(int16_t* p, uint4 v) {
  const w_u32x4 x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<w_u32x4*>(p));
}
__device__ __forceinline__ void w_unpack(uint4 lo, uint4 hi, float (&c)[16]) {
  float a[8], b[8];
  s16x8_to_f(lo, a);
  s16x8_to_f(hi, b);
#pragma unroll
  for (int k = 0; k < 8; k++) { c[k] = a[k]; c[k + 8] = b[k]; }
}
// any |c| > bound in this wavefront?
__device__ __forceinline__ bool w_any_above(const float (&c)[16], float bound) {
  float m = fmaxf(fabsf(c[0]), fabsf(c[1]));
#pragma unroll
  for (int k = 2; k < 16; k += 2) m = fmaxf(m, fmaxf(fabsf(c[k]), fabsf(c[k + 1])));
  return __any(m > bound);
}
// integer-valued |c| < 2^15 -> int16, S order
__device__ __forceinline__ void w_pack(const float (&c)[16], uint4& lo, uint4& hi) {
  float a[8], b[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { a[k] = c[k]; b[k] = c[k + 8]; }
  lo = f_to_s16x8(a);
  hi = f_to_s16x8(b);
}

// ---------------------------------------------------------------- kernels
// One launch-sized grid: a wavefront takes 4 U consecutive polynomials, requests all of them, then
// transforms and stores them four at a time and ends -- workgroups are dispatched in index order, so
// the chip sweeps the arrays front to back as one moving window (grid-stride loops and per-workgroup
// runs spread the window and measured 15-20 % less out of HBM, tools/microbench/stream_pattern.hip).
// In-place use is safe: a wavefront reads all of its polynomials before it writes any.
enum WOp { W_NTT = 0, W_NTT_CANON, W_INVNTT_CANON, W_BASEMUL_CANON, W_MUL_CANON };

// constants of the final scalings (centred residues mod q)
constexpr float W_MONT_INV = 169.0f;          // 2^-16 mod q
constexpr float W_INV_TOMONT = 512.0f;        // 2^16 / 128 mod q: what invntt_tomont leaves on its result (ref/ntt.c:123-125)
constexpr float W_INV_TOMONT_Z1 = -266.0f;    // 512 * zeta[1] mod q = 512 * -1600
constexpr float W_INV_PLAIN = KY_NINV_PLAIN;  // 128^-1 = -26
constexpr float W_INV_PLAIN_Z1 = 1652.0f;     // -26 * -1600 mod q

template <int OP>
struct WTwiddles {
  WTwFwd F;
  __device__ __forceinline__ void load(int l) { w_tw_fwd(F, l); }
};
template <>
struct WTwiddles<W_INVNTT_CANON> {
  WTwInv I;
  __device__ __forceinline__ void load(int l) { w_tw_inv(I, l); }
};

// four polynomials of the wavefront: packed inputs -> packed result
template <int OP>
__device__ __forceinline__ void w_op(const WLane& L, const WTwiddles<OP>& T, uint4 a0, uint4 a1, uint4 b0, uint4 b1, uint4& lo, uint4& hi) {
  float c[16];
  if constexpr (OP == W_NTT || OP == W_NTT_CANON) {
    w_unpack(a0, a1, c);
    w_centre_if(w_any_above(c, 4095.0f), c);
    w_ntt_fwd(L, T.F, c);
    if constexpr (OP == W_NTT) w_fred16(c);  // the Barrett-centred representative of ref/poly.c:244-248
    else w_fred16_canon(c);
  } else if constexpr (OP == W_INVNTT_CANON) {
    w_unpack(a0, a1, c);
    w_centre_if(w_any_above(c, 1665.0f), c);
    w_ntt_inv(L, T.I, c, W_INV_TOMONT, W_INV_TOMONT_Z1);
  } else if constexpr (OP == W_BASEMUL_CANON) {
    // poly_basemul_montgomery carries 2^-16: it goes onto b, which has to be centred anyway
    float x[16], y[16];
    w_unpack(a0, a1, x);
    w_unpack(b0, b1, y);
    w_centre_if(w_any_above(x, 4095.0f), x);  // against |y| <= 1665 below: two products <= 13.7 M
    w_centre_if(w_any_above(y, 4095.0f), y);
    const Tw minv = mk_tw(W_MONT_INV);
#pragma unroll
    for (int k = 0; k < 16; k++) y[k] = fmulc(y[k], minv);  // 169 * 4095 < 2^24
    w_basemul(c, x, y, T.F);
    w_fred16_canon(c);
  } else {
    // invntt_tomont(poly_ntt(a) o poly_ntt(b)): the 2^-16 of the product and the 2^16 of the inverse
    // transform cancel, what is left is the plain product of a and b in Z_q[X]/(X^256+1)
    float x[16], y[16];
    w_unpack(a0, a1, x);
    w_centre_if(w_any_above(x, 4095.0f), x);
    w_ntt_fwd(L, T.F, x);
    w_fred16(x);
    w_unpack(b0, b1, y);
    w_centre_if(w_any_above(y, 4095.0f), y);
    w_ntt_fwd(L, T.F, y);  // left at <= 8325
    w_basemul(c, x, y, T.F);
    w_fred16(c);
    WTwInv TI;
    w_tw_inv(TI, L.l);
    w_ntt_inv(L, TI, c, W_INV_PLAIN, W_INV_PLAIN_Z1);
  }
  w_pack(c, lo, hi);
}

// U batches of four polynomials per wavefront: 16 U polynomials per 256-lane workgroup
template <int OP, int U, int WAVES = 1>
__global__ void __launch_bounds__(256, WAVES) k_wpoly(long npoly, int16_t* r, const int16_t* a, const int16_t* b) {
  constexpr bool TWO = OP == W_BASEMUL_CANON || OP == W_MUL_CANON;
  __shared__ __attribute__((aligned(16))) float wk[4][1024];
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const WLane L = w_lane(wk[wave]);
  const long p0 = ((long)blockIdx.x * 4 + wave) * (4 * U) + (lane >> 4);
  uint4 a0[U], a1[U], b0[U], b1[U];
  long off[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const long pi = p0 + 4 * u;
    off[u] = (pi < npoly ? pi : npoly - 1) * 256 + 8 * L.l;  // surplus groups of the last workgroup redo the last polynomial
    a0[u] = w_ld16(a + off[u]);
    a1[u] = w_ld16(a + off[u] + 128);
    if constexpr (TWO) {
      b0[u] = w_ld16(b + off[u]);
      b1[u] = w_ld16(b + off[u] + 128);
    } else {
      b0[u] = b1[u] = make_uint4(0, 0, 0, 0);
    }
  }
  WTwiddles<OP> T;
  T.load(L.l);
#pragma unroll
  for (int u = 0; u < U; u++) {
    uint4 lo, hi;
    w_op<OP>(L, T, a0[u], a1[u], b0[u], b1[u], lo, hi);
    if (p0 + 4 * u < npoly) {
      w_st16(r + off[u], lo);
      w_st16(r + off[u] + 128, hi);
    }
  }
}

}  // namespace ky
