// fpoly.h -- Z_3329[X]/(X^256+1) arithmetic on the fp32 pipe.
//
// Measured on MI355X (profiles/r01_microbench_valu_rate_*.txt): v_fma_f32 /
// v_mul_f32 / v_add_f32 issue at full rate, while every integer multiply
// (v_mul_i32_i24, v_mad_i32_i24, v_mul_lo_u32, v_mad_i32_i16) is half rate.
// All values on this path are integers below 2^24 in magnitude, which fp32
// represents exactly, so a modular product is
//     p = a*b              (exact while |a*b| < 2^24)
//     k = rint(p * (1/q))  (may be off by one near a half; harmless)
//     r = fma(-k, q, p)    (exact: an integer with |r| <= q/2 + 3)
// i.e. four full-rate ops instead of a Montgomery product's four half-rate
// ones, and there is no Montgomery domain at all: coefficients are plain
// residues, twiddles are 17^brv7(i) mod q, the inverse transform ends with
// 128^-1.  The bytes the KEM emits are canonical residues (every serializer of
// the reference canonicalises: ref/poly.c:118-121,28-30,188-190;
// ref/polyvec.c:24-26), so they do not depend on the representative or on the
// Montgomery factors the reference carries internally; tests pin that with
// SHA256SUMS through this path.
//
// Magnitude discipline (products must stay below 2^24 = 16.7 M):
//   |twiddle| <= 1664.  A value that enters a product is kept <= 10 000.
//   fred() output: |r| <= 1668.  Bounds are stated at each call site.
//
// Same 32-work-items-per-polynomial layout and pass structure as poly.h; LDS
// buffers hold fp32 with an XOR swizzle so that all three access maps are
// bank-conflict-free.
#pragma once
#include "poly.h"

namespace ky {

__constant__ __attribute__((aligned(16))) const float ZETAF[128] = KY_ZETAS_PLAIN_INIT;  // aligned: wpoly.h reads runs of 2 and 4

constexpr float QF = 3329.0f;
constexpr float QINVF = 1.0f / 3329.0f;

// centred residue of an integer-valued x, |x| < 2^24.  rint() is half rate on
// gfx950, so the quotient is rounded by adding and subtracting 1.5 * 2^23
// (three full-rate ops in all).
constexpr float RND_MAGIC = 12582912.0f;
__device__ __forceinline__ float fred(float x) {
  float k = __builtin_fmaf(x, QINVF, RND_MAGIC) - RND_MAGIC;
  return __builtin_fmaf(k, -QF, x);
}
__device__ __forceinline__ float fmm(float a, float b) { return fred(a * b); }

// [0,q) integer of a centred residue (|r| < q)
__device__ __forceinline__ int fcanon(float r) {
  int v = (int)r;
  return v + ((v >> 31) & Q);
}

// round(x * 2^D / q) mod 2^D of an integer-valued x, |x| <= 5000, ANY
// representative of its residue class -- the reference's compression
// ((u << D) + q/2) / q & mask of the canonical u (ref/poly.c:30,45,
// ref/polyvec.c:26,50; D = 1 is poly_tomsg, ref/poly.c:187-190).  One FMA puts
// x * (2^D/q) + 1.5*2^23 into a float whose low mantissa bits ARE the rounded
// quotient in two's complement, so the result is a mask away.  Exact because
// x * 2^D / q is never within 1/(2q) = 1.5e-4 of a rounding boundary (q is
// odd) while the error of the rounded constant is <= |x| * 2^-25 < 1.5e-4;
// different representatives differ by exact multiples of 2^D.  Checked
// exhaustively in tests/test_host_logic.py.
template <int D>
__device__ __forceinline__ unsigned fcompress(float x) {
  constexpr float C = (float)(1 << D) / 3329.0f;
  return __float_as_uint(__builtin_fmaf(x, C, RND_MAGIC)) & ((1u << D) - 1);
}

// ---------------------------------------------------------------- LDS maps (fp32, swizzled)
// element i lives at i ^ ((i >> 5) << 2): STRIDE, MIDDLE and CONTIG accesses
// then each touch 32 distinct banks per 32-lane group.
__device__ __forceinline__ void f_put_stride(float* p, int t, const float (&c)[8]) {
#pragma unroll
  for (int m = 0; m < 8; m++) p[32 * m + (t ^ (4 * m))] = c[m];
}
__device__ __forceinline__ void f_get_stride(const float* p, int t, float (&c)[8]) {
#pragma unroll
  for (int m = 0; m < 8; m++) c[m] = p[32 * m + (t ^ (4 * m))];
}
__device__ __forceinline__ void f_put_middle(float* p, int t, const float (&c)[8]) {
  const int hi = t >> 2, base = 32 * hi + (t & 3);
#pragma unroll
  for (int m = 0; m < 8; m++) p[base + 4 * (m ^ hi)] = c[m];
}
__device__ __forceinline__ void f_get_middle(const float* p, int t, float (&c)[8]) {
  const int hi = t >> 2, base = 32 * hi + (t & 3);
#pragma unroll
  for (int m = 0; m < 8; m++) c[m] = p[base + 4 * (m ^ hi)];
}
// coefficients 8t..8t+7: two aligned quads, each moved as a unit by the swizzle
__device__ __forceinline__ void f_put_contig(float* p, int t, const float (&c)[8]) {
  const int s = (t >> 2) << 2;
  *reinterpret_cast<float4*>(p + ((8 * t) ^ s)) = make_float4(c[0], c[1], c[2], c[3]);
  *reinterpret_cast<float4*>(p + ((8 * t + 4) ^ s)) = make_float4(c[4], c[5], c[6], c[7]);
}
__device__ __forceinline__ void f_get_contig(const float* p, int t, float (&c)[8]) {
  const int s = (t >> 2) << 2;
  float4 a = *reinterpret_cast<const float4*>(p + ((8 * t) ^ s));
  float4 b = *reinterpret_cast<const float4*>(p + ((8 * t + 4) ^ s));
  c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
}
// unswizzled fp32 polynomial storage that is only ever read CONTIG
__device__ __forceinline__ void f_put_plain(float* p, int t, const float (&c)[8]) {
  *reinterpret_cast<float4*>(p + 8 * t) = make_float4(c[0], c[1], c[2], c[3]);
  *reinterpret_cast<float4*>(p + 8 * t + 4) = make_float4(c[4], c[5], c[6], c[7]);
}
__device__ __forceinline__ void f_get_plain(const float* p, int t, float (&c)[8]) {
  float4 a = *reinterpret_cast<const float4*>(p + 8 * t), b = *reinterpret_cast<const float4*>(p + 8 * t + 4);
  c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
}

// Twiddles indexed per lane come from a 512-byte LDS copy (Z): a constant-memory
// access with a per-lane index is a vector memory load whose latency would be
// exposed three times per transform.  Compile-time indices stay on ZETAF
// (scalar loads).  Call once per workgroup, ends with a barrier.
__device__ __forceinline__ void f_twiddles_to_lds(float* Z) {
  if (threadIdx.x < 128) Z[threadIdx.x] = ZETAF[threadIdx.x];
  __syncthreads();
}

// ---------------------------------------------------------------- butterflies
// z * b mod q as three FMAs (wpoly.h's fmulc; -DKY_FMULC=1 builds the 32-lane engine on it for A/Bs):
// k' = fma(b, z/q, M), m = fma(k', -q, q M) = -q rint(b z / q), r = fma(z, b, m).  Same bound as fmm
// (|z b| < 2^24), |r| <= 1665.
#ifndef KY_FMULC
#define KY_FMULC 0
#endif
__device__ __forceinline__ float fmm_c(float z, float b) {
#if KY_FMULC
  const float k = __builtin_fmaf(b, z * QINVF, RND_MAGIC);
  const float m = __builtin_fmaf(k, -QF, QF * RND_MAGIC);
  return __builtin_fmaf(z, b, m);
#else
  return fmm(z, b);
#endif
}
__device__ __forceinline__ void f_ct(float& a, float& b, float z) {  // Cooley-Tukey, ref/ntt.c:89-91
  float t = fmm_c(z, b);
  b = a - t;
  a = a + t;
}
__device__ __forceinline__ void f_gs(float& a, float& b, float z) {  // Gentleman-Sande, ref/ntt.c:116-119 (sum left unreduced)
  float t = a;
  a = t + b;
  b = fmm_c(z, b - t);
}

// ---------------------------------------------------------------- forward NTT
// in : c[] CONTIG, |c| <= 4095 (noise, decompressed or 12-bit unpacked values)
// out: c[] CONTIG, centred residues of the plain NTT (= ref/poly.c:244-248 mod q)
// p  : the group's swizzled fp32 scratch buffer (256 floats)
// bounds: each layer adds <= 1668, so 4095 -> 9099 after three layers (products
// <= 1664 * 7431 = 12.4 M); reduce; 1668 -> 6672 after three more (products
// <= 8.4 M); last layer multiplies values <= 6672 (11.2 M).
// SMALL_IN: |c| <= 78 on entry (noise polynomials, also scaled by 128^-1 = -26).
// Values then stay <= 78 + 6 * 1665 = 10068 up to the last layer, whose products
// are <= 1664 * 10068 = 16.75 M < 2^24, so the reduction after the third layer
// is left out.
// KY_ABLATE_NTT_LDS (tools/microbench only): the transposes between the passes are left out, which
// makes the results wrong and prices what the LDS round trips cost the arithmetic kernels.
#ifdef KY_ABLATE_NTT_LDS
#define KY_XPOSE(put, get) do { } while (0)
#else
#define KY_XPOSE(put, get) do { KY_WSYNC(); put(p, t, c); KY_WSYNC(); get(p, t, c); } while (0)
#endif
template <bool SMALL_IN = false>
__device__ __forceinline__ void f_ntt_fwd(float* p, const float* Z, int t, float (&c)[8]) {
  KY_XPOSE(f_put_contig, f_get_stride);
#pragma unroll
  for (int m = 0; m < 4; m++) f_ct(c[m], c[m + 4], ZETAF[1]);
  f_ct(c[0], c[2], ZETAF[2]); f_ct(c[1], c[3], ZETAF[2]);
  f_ct(c[4], c[6], ZETAF[3]); f_ct(c[5], c[7], ZETAF[3]);
  f_ct(c[0], c[1], ZETAF[4]); f_ct(c[2], c[3], ZETAF[5]);
  f_ct(c[4], c[5], ZETAF[6]); f_ct(c[6], c[7], ZETAF[7]);
  if constexpr (!SMALL_IN) {
#pragma unroll
    for (int m = 0; m < 8; m++) c[m] = fred(c[m]);
  }
  KY_XPOSE(f_put_stride, f_get_middle);
  {
    const int hi = t >> 2;
    float z = Z[8 + hi];
#pragma unroll
    for (int m = 0; m < 4; m++) f_ct(c[m], c[m + 4], z);
    float z0 = Z[16 + 2 * hi], z1 = Z[17 + 2 * hi];
    f_ct(c[0], c[2], z0); f_ct(c[1], c[3], z0);
    f_ct(c[4], c[6], z1); f_ct(c[5], c[7], z1);
    f_ct(c[0], c[1], Z[32 + 4 * hi]); f_ct(c[2], c[3], Z[33 + 4 * hi]);
    f_ct(c[4], c[5], Z[34 + 4 * hi]); f_ct(c[6], c[7], Z[35 + 4 * hi]);
  }
  KY_XPOSE(f_put_middle, f_get_contig);
  {
    float z0 = Z[64 + 2 * t], z1 = Z[65 + 2 * t];
    f_ct(c[0], c[2], z0); f_ct(c[1], c[3], z0);
    f_ct(c[4], c[6], z1); f_ct(c[5], c[7], z1);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = fred(c[k]);
}

// ---------------------------------------------------------------- inverse NTT
// in : c[] CONTIG, |c| <= 1668 (NTT domain)
// out: c[] CONTIG, centred residues of the plain inverse NTT (no Montgomery
//      factor: it replaces invntt_tomont applied to a basemul result, whose
//      2^-16 and 2^16 cancel in the reference, ref/ntt.c:106-126)
// bounds: sums double per layer and are reduced every second layer, so a
// difference entering a product is <= 2 * 3336 = 6672 (11.2 M).
// SCALE = false leaves out the final multiplication by 128^-1 (the caller has
// folded it into a linear factor upstream, e.g. into r^): outputs are then
// |c| <= 3330 (sums of two reduced values) instead of <= 1665.
template <bool SCALE = true>
__device__ __forceinline__ void f_ntt_inv(float* p, const float* Z, int t, float (&c)[8]) {
  {
    float z0 = Z[127 - 2 * t], z1 = Z[126 - 2 * t];
    f_gs(c[0], c[2], z0); f_gs(c[1], c[3], z0);
    f_gs(c[4], c[6], z1); f_gs(c[5], c[7], z1);
  }
  KY_XPOSE(f_put_contig, f_get_middle);
  {
    const int hi = t >> 2;
    f_gs(c[0], c[1], Z[63 - 4 * hi]); f_gs(c[2], c[3], Z[62 - 4 * hi]);
    f_gs(c[4], c[5], Z[61 - 4 * hi]); f_gs(c[6], c[7], Z[60 - 4 * hi]);
    c[0] = fred(c[0]); c[2] = fred(c[2]); c[4] = fred(c[4]); c[6] = fred(c[6]);  // sums of two layers
    float za = Z[31 - 2 * hi], zb = Z[30 - 2 * hi];
    f_gs(c[0], c[2], za); f_gs(c[1], c[3], za);
    f_gs(c[4], c[6], zb); f_gs(c[5], c[7], zb);
    float z = Z[15 - hi];
#pragma unroll
    for (int m = 0; m < 4; m++) f_gs(c[m], c[m + 4], z);
    c[0] = fred(c[0]); c[1] = fred(c[1]); c[2] = fred(c[2]); c[3] = fred(c[3]);
  }
  KY_XPOSE(f_put_middle, f_get_stride);
  f_gs(c[0], c[1], ZETAF[7]); f_gs(c[2], c[3], ZETAF[6]);
  f_gs(c[4], c[5], ZETAF[5]); f_gs(c[6], c[7], ZETAF[4]);
  f_gs(c[0], c[2], ZETAF[3]); f_gs(c[1], c[3], ZETAF[3]);
  f_gs(c[4], c[6], ZETAF[2]); f_gs(c[5], c[7], ZETAF[2]);
  c[0] = fred(c[0]); c[1] = fred(c[1]); c[4] = fred(c[4]); c[5] = fred(c[5]);
#pragma unroll
  for (int m = 0; m < 4; m++) f_gs(c[m], c[m + 4], ZETAF[1]);
  if constexpr (SCALE) {
#pragma unroll
    for (int m = 0; m < 8; m++) c[m] = fmm(c[m], KY_NINV_PLAIN);  // |c| <= 3336, * 26
  }
  KY_XPOSE(f_put_stride, f_get_contig);
}

// ---------------------------------------------------------------- basemul
// acc[] += a[] o b[] (plain product in Z_q[X]/(X^2 -+ zeta)), each term reduced
// before it is added: |a| <= 4095 (matrix entries, unpacked key bytes),
// |b| <= 1668 (NTT output), so a single product is <= 6.9 M and a two-product
// sum <= 13.7 M.  ref/ntt.c:139-146, ref/poly.c:273-280, ref/polyvec.c:194-198.
__device__ __forceinline__ void f_basemul_acc(float (&acc)[8], const float (&a)[8], const float (&b)[8], const float* Z, int t) {
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const float z = Z[64 + 2 * t + g];
    const int o = 4 * g;
    float h0 = fmm(a[o + 1], b[o + 1]), h1 = fmm(a[o + 3], b[o + 3]);
    acc[o + 0] += fred(__builtin_fmaf(h0, z, a[o + 0] * b[o + 0]));
    acc[o + 1] += fred(__builtin_fmaf(a[o + 0], b[o + 1], a[o + 1] * b[o + 0]));
    acc[o + 2] += fred(__builtin_fmaf(h1, -z, a[o + 2] * b[o + 2]));
    acc[o + 3] += fred(__builtin_fmaf(a[o + 2], b[o + 3], a[o + 3] * b[o + 2]));
  }
}

// ---------------------------------------------------------------- cached multiplicand
// A polynomial b (NTT domain, centred, |b| <= 1665) that multiplies several
// others -- r^ against the K+1 rows of encryption, s^ against the K rows of
// key generation -- is kept per work-item as its 8 coefficients plus the four
// "twisted" odd ones zb[p] = (+-zeta) * b[2p+1] mod q.  A base multiplication is
// then pure multiply-add (ref/ntt.c:139-146 with the zeta product hoisted):
//     c[2p]   += a[2p] * b[2p]   + a[2p+1] * zb[p]
//     c[2p+1] += a[2p] * b[2p+1] + a[2p+1] * b[2p]
// four FMAs per coefficient pair instead of five modular products, and the
// sums over j stay UNREDUCED, which is exact while they stay below 2^24:
//   * operands centred, |a| <= 1665 (f_mac after fred): six products <= 16.64 M.
//   * matrix entries are canonical, a in [0, 3328].  Write a = a' + 1664 with
//     |a'| <= 1664: sum a*b = sum a'*b + 1664*S, S = sum of the b-side factors
//     (known once b is known, the same for every row).  The accumulator starts
//     at corr = -q * rne(1664*S / q), a multiple of q within 1665 of -1664*S, so
//     every partial sum  corr + sum_{first t} a*b  =  sum_{first t} a'*b
//     - 1664*sum_{rest} b + (corr + 1664*S)  is bounded by 1664*1665*T + 1665
//     for T products in all: T = 6 -> 16.63 M < 2^24 = 16.78 M.  The result is
//     congruent to sum a*b and costs nothing per row.
// So at most three polynomials (six products per coefficient) share one
// accumulation; Kyber1024 splits its four into two groups (FC_GROUP).
struct FCache {
  float b[8], zb[4];
};
__device__ __forceinline__ void f_cache_make(FCache& c, const float (&b)[8], const float* Z, int t) {
  const float z0 = Z[64 + 2 * t], z1 = Z[65 + 2 * t];
#pragma unroll
  for (int k = 0; k < 8; k++) c.b[k] = b[k];
  c.zb[0] = fmm(z0, b[1]);
  c.zb[1] = fmm(-z0, b[3]);
  c.zb[2] = fmm(z1, b[5]);
  c.zb[3] = fmm(-z1, b[7]);
}
// S[] += the b-side factor sums of one cached polynomial
__device__ __forceinline__ void f_cache_sum(float (&S)[8], const FCache& c) {
#pragma unroll
  for (int p = 0; p < 4; p++) {
    S[2 * p] += c.b[2 * p] + c.zb[p];
    S[2 * p + 1] += c.b[2 * p + 1] + c.b[2 * p];
  }
}
constexpr float FC_OFFSET = 1664.0f;
// accumulator start value for canonical a (see above); |S| <= 6 * 1665
__device__ __forceinline__ float f_corr(float S) {
  const float x = FC_OFFSET * S;  // exact, < 2^24
  const float k = __builtin_fmaf(x, QINVF, RND_MAGIC) - RND_MAGIC;
  return -QF * k;
}
// acc[] += a[] o c, no reduction
__device__ __forceinline__ void f_mac(float (&acc)[8], const float (&a)[8], const FCache& c) {
#pragma unroll
  for (int p = 0; p < 4; p++) {
    acc[2 * p] = __builtin_fmaf(a[2 * p], c.b[2 * p], __builtin_fmaf(a[2 * p + 1], c.zb[p], acc[2 * p]));
    acc[2 * p + 1] = __builtin_fmaf(a[2 * p], c.b[2 * p + 1], __builtin_fmaf(a[2 * p + 1], c.b[2 * p], acc[2 * p + 1]));
  }
}
// 8 packed unsigned 16-bit values -> fp32 (one v_cvt_f32_u32 with a sub-dword select each)
__device__ __forceinline__ void u16x8_to_f(uint4 v, float (&f)[8]) {
  f[0] = (float)(v.x & 0xFFFFu); f[1] = (float)(v.x >> 16);
  f[2] = (float)(v.y & 0xFFFFu); f[3] = (float)(v.y >> 16);
  f[4] = (float)(v.z & 0xFFFFu); f[5] = (float)(v.z >> 16);
  f[6] = (float)(v.w & 0xFFFFu); f[7] = (float)(v.w >> 16);
}

// compact persistent storage: centred residues as int16, CONTIG access only
__device__ __forceinline__ void f_put_i16(int16_t* p, int t, const float (&c)[8]) {
  int v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = (int)c[k];
  lds_put_contig(p, t, v);
}
__device__ __forceinline__ void f_get_i16(const int16_t* p, int t, float (&c)[8]) {
  int v[8];
  lds_get_contig(p, t, v);
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = (float)v[k];
}

// 8 packed signed 16-bit values -> fp32 (one v_cvt_f32_i32 with a sign-extending sub-dword select each)
__device__ __forceinline__ void s16x8_to_f(uint4 v, float (&f)[8]) {
  f[0] = (float)(short)v.x; f[1] = (float)((int)v.x >> 16);
  f[2] = (float)(short)v.y; f[3] = (float)((int)v.y >> 16);
  f[4] = (float)(short)v.z; f[5] = (float)((int)v.z >> 16);
  f[6] = (float)(short)v.w; f[7] = (float)((int)v.w >> 16);
}
// integer-valued floats, |c| < 2^15 -> 8 packed int16: adding 1.5 * 2^23 leaves the two's
// complement value in the low mantissa bits (full-rate add instead of the half-rate
// v_cvt_i32_f32), one v_perm_b32 joins two low halves
__device__ __forceinline__ uint4 f_to_s16x8(const float (&c)[8]) {
  uint32_t u[8];
#pragma unroll
  for (int k = 0; k < 8; k++) u[k] = __float_as_uint(c[k] + RND_MAGIC);
  return make_uint4(__builtin_amdgcn_perm(u[1], u[0], 0x05040100u), __builtin_amdgcn_perm(u[3], u[2], 0x05040100u),
                    __builtin_amdgcn_perm(u[5], u[4], 0x05040100u), __builtin_amdgcn_perm(u[7], u[6], 0x05040100u));
}

__device__ __forceinline__ void i2f8(const int (&c)[8], float (&f)[8]) {
#pragma unroll
  for (int k = 0; k < 8; k++) f[k] = (float)c[k];
}
__device__ __forceinline__ void f2canon8(const float (&f)[8], int (&c)[8]) {
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = fcanon(f[k]);
}

}  // namespace ky
