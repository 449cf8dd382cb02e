// wpoly_dpp.hip -- north_star's "final layers in-register with DPP / ds_swizzle", built and timed
// against the product's LDS-transposed wide engine (kyber_amd/csrc/wpoly.h) on 2^20 polynomials.
//
// Same shape as wpoly.h (16 work-items per polynomial, 16 coefficients each, S layout: local index
// bits 7,2,1,0), but the four layers on index bits 6..3 never leave the lanes' registers: a lane
// multiplies its own value (by -zeta where it holds the butterfly's second input, by 1 where it
// holds the first: the product reduces the value either way), then adds or subtracts its partner's
// product, read through the DPP crossbar by one v_fmac_f32_dpp:
//     first input's lane :  a' = a + zeta b = d - d_partner        d = a,        d_partner = -zeta b
//     second input's lane:  b' = a - zeta b = d + d_partner        d = -zeta b,  d_partner = a
// i.e. c = fma(d_partner, sigma, d) with a per-lane sign.  gfx9-class DPP reaches the partners
// l^8 (row_ror:8), l^7 (row_half_mirror), l^2 and l^1 (quad_perm) in one instruction each, so the lane
// number is the linear image l = 8 b6 ^ 7 b5 ^ 2 b4 ^ b3 of the index bits; memory is addressed
// through the inverse map.  No LDS, no mid-transform reduction (every cross-lane layer reduces),
// 4 instructions per coefficient and cross-lane layer against 2.5 in registers.
//
// Both directions are built: forward (ref/ntt.c:80-95) and inverse (ref/ntt.c:106-126, GS: e = c -+
// c_partner, then the product), each verified mod q against an integer model of the reference's loops,
// and the fused product ntt, ntt, basemul, invntt on them.
#include "../../kyber_amd/csrc/arith_kernels.h"
#include "../../kyber_amd/csrc/wpoly.h"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using namespace ky;

static const int ZM[128] = KY_ZETAS_INIT;
static long md(long x) { x %= 3329; return x < 0 ? x + 3329 : x; }
static long fq(long a, long b) { return md(md(a * b) * 169); }
static void m_ntt(long* r) {
  int k = 1;
  for (int len = 128; len >= 2; len >>= 1)
    for (int start = 0; start < 256; start += 2 * len) {
      long z = ZM[k++];
      for (int j = start; j < start + len; j++) { long t = fq(z, r[j + len]); r[j + len] = md(r[j] - t); r[j] = md(r[j] + t); }
    }
}
static void m_invntt_tomont(long* r) {
  int k = 127;
  for (int len = 2; len <= 128; len <<= 1)
    for (int start = 0; start < 256; start += 2 * len) {
      long z = ZM[k--];
      for (int j = start; j < start + len; j++) { long t = r[j]; r[j] = md(t + r[j + len]); r[j + len] = fq(z, r[j + len] - t); }
    }
  for (int j = 0; j < 256; j++) r[j] = fq(r[j], 1441);
}
static void m_basemul(long* r, const long* a, const long* b) {
  for (int i = 0; i < 64; i++)
    for (int s = 0; s < 2; s++) {
      const long z = s ? -ZM[64 + i] : ZM[64 + i];
      const long *x = a + 4 * i + 2 * s, *y = b + 4 * i + 2 * s;
      r[4 * i + 2 * s] = md(fq(fq(x[1], y[1]), z) + fq(x[0], y[0]));
      r[4 * i + 2 * s + 1] = md(fq(x[0], y[1]) + fq(x[1], y[0]));
    }
}

// ---------------------------------------------------------------- cross-lane pieces
constexpr int DPP_XOR8 = 0x128;  // row_ror:8
constexpr int DPP_XOR7 = 0x141;  // row_half_mirror
constexpr int DPP_XOR2 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int DPP_XOR1 = 0xB1;   // quad_perm:[1,0,3,2]
template <int CTRL>
__device__ __forceinline__ float dpp_get(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
struct XTw {
  Tw w[2];      // by the local h: -zeta (second input's lane) or 1 (first input's lane)
  float sigma;  // -1 on the first input's lane, +1 on the second's
};
// c[k] = c_partner[k] * sigma + c[k] for the 16 values of a lane: one v_fmac_f32_dpp each.  The compiler
// keeps update_dpp as v_mov_b32_dpp plus a separate FMA (5 instructions per coefficient and layer,
// -DKY_DPP_MOV builds that form); the block below is the folded form.  s_nop 1: a DPP read needs two
// wait states after the VALU write of its source, which the compiler does not track into inline asm.
#define X_F(i, ctrl) "v_fmac_f32_dpp %" #i ", %" #i ", %16 " ctrl " row_mask:0xf bank_mask:0xf\n"
#define X_FMAC16(c, s, ctrl)                                                                                        \
  asm("s_nop 1\n" X_F(0, ctrl) X_F(1, ctrl) X_F(2, ctrl) X_F(3, ctrl) X_F(4, ctrl) X_F(5, ctrl) X_F(6, ctrl) X_F(7, ctrl)  \
      X_F(8, ctrl) X_F(9, ctrl) X_F(10, ctrl) X_F(11, ctrl) X_F(12, ctrl) X_F(13, ctrl) X_F(14, ctrl) X_F(15, ctrl)       \
      : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]),  \
        "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15])                     \
      : "v"(s))
template <int CTRL>
__device__ __forceinline__ void x_combine(float (&c)[16], float sigma) {
#ifdef KY_DPP_MOV
#pragma unroll
  for (int k = 0; k < 16; k++) c[k] = __builtin_fmaf(dpp_get<CTRL>(c[k]), sigma, c[k]);
#else
  if constexpr (CTRL == DPP_XOR8) X_FMAC16(c, sigma, "row_ror:8");
  else if constexpr (CTRL == DPP_XOR7) X_FMAC16(c, sigma, "row_half_mirror");
  else if constexpr (CTRL == DPP_XOR2) X_FMAC16(c, sigma, "quad_perm:[2,3,0,1]");
  else X_FMAC16(c, sigma, "quad_perm:[1,0,3,2]");
#endif
}
// forward layer: d = w c (reduced), c = d + sigma d_partner.  in: |c| w < 2^24; out: |c| <= 3330
template <int CTRL>
__device__ __forceinline__ void x_ct(float (&c)[16], const XTw& t) {
#pragma unroll
  for (int k = 0; k < 16; k++) c[k] = fmulc(c[k], t.w[k >> 3]);
  x_combine<CTRL>(c, t.sigma);
}
// inverse layer (GS): e = c + sigma c_partner (first lane: a + b; second: b - a), c = w e with w = 1 / zeta.
// in: |c| <= 3330 -> |e| <= 6660; out: |c| <= 1665
template <int CTRL>
__device__ __forceinline__ void x_gs(float (&c)[16], const XTw& t) {
  x_combine<CTRL>(c, t.sigma);
#pragma unroll
  for (int k = 0; k < 16; k++) c[k] = fmulc(c[k], t.w[k >> 3]);
}

struct XLane {
  int m;  // coefficient group: this lane holds coefficients 128 h + 8 m + k
  XTw f2, f3, f4, f5;  // forward, index bits 6, 5, 4, 3
  XTw i5, i4, i3, i2;  // inverse, index bits 3, 4, 5, 6
  Tw fz4[2], fz2[4], iz2[4], iz4[2];
};
__device__ __forceinline__ XTw x_tw(float z0, float z1, int second, float sig_first) {
  XTw t;
  t.w[0] = mk_tw(second ? z0 : 1.0f);
  t.w[1] = mk_tw(second ? z1 : 1.0f);
  t.sigma = second ? -sig_first : sig_first;
  return t;
}
__device__ __forceinline__ void x_lane(XLane& X, int l) {
  const int b6 = (l >> 3) & 1, b5 = (l >> 2) & 1, b4 = ((l >> 1) ^ (l >> 2)) & 1, b3 = (l ^ (l >> 2)) & 1;
  const int m = 8 * b6 + 4 * b5 + 2 * b4 + b3;
  X.m = m;
  // forward twiddle index = 2^layer + (coefficient index >> (8 - layer)), ref/ntt.c:84-86; the second
  // input's lane multiplies by -zeta and then ADDS the first's value
  X.f2 = x_tw(-ZETAF[2], -ZETAF[3], b6, -1.0f);
  X.f3 = x_tw(-ZETAF[4 + b6], -ZETAF[6 + b6], b5, -1.0f);
  X.f4 = x_tw(-ZETAF[8 + (m >> 2)], -ZETAF[12 + (m >> 2)], b4, -1.0f);
  X.f5 = x_tw(-ZETAF[16 + (m >> 1)], -ZETAF[24 + (m >> 1)], b3, -1.0f);
  X.fz4[0] = mk_tw(ZETAF[32 + m]); X.fz4[1] = mk_tw(ZETAF[48 + m]);
  X.fz2[0] = mk_tw(ZETAF[64 + 2 * m]); X.fz2[1] = mk_tw(ZETAF[65 + 2 * m]);
  X.fz2[2] = mk_tw(ZETAF[96 + 2 * m]); X.fz2[3] = mk_tw(ZETAF[97 + 2 * m]);
  // inverse twiddle index = 2^(layer+1) - 1 - (coefficient index >> (8 - layer)), ref/ntt.c:110-113;
  // the first lane takes a + b (sigma +1, w = 1), the second (b - a) zeta (sigma -1)
  X.iz2[0] = mk_tw(ZETAF[127 - 2 * m]); X.iz2[1] = mk_tw(ZETAF[126 - 2 * m]);
  X.iz2[2] = mk_tw(ZETAF[95 - 2 * m]); X.iz2[3] = mk_tw(ZETAF[94 - 2 * m]);
  X.iz4[0] = mk_tw(ZETAF[63 - m]); X.iz4[1] = mk_tw(ZETAF[47 - m]);
  X.i5 = x_tw(ZETAF[31 - (m >> 1)], ZETAF[23 - (m >> 1)], b3, 1.0f);
  X.i4 = x_tw(ZETAF[15 - (m >> 2)], ZETAF[11 - (m >> 2)], b4, 1.0f);
  X.i3 = x_tw(ZETAF[7 - b6], ZETAF[5 - b6], b5, 1.0f);
  X.i2 = x_tw(ZETAF[3], ZETAF[2], b6, 1.0f);
}

// in: |c| <= 4095, S layout by m; out: NTT, |c| <= 6660, not reduced
__device__ __forceinline__ void x_ntt_fwd(const XLane& X, float (&c)[16]) {
  const Tw z128 = mk_tw(ZETAF[1]);
#pragma unroll
  for (int k = 0; k < 8; k++) w_ct(c[k], c[k + 8], z128);  // <= 5760
  x_ct<DPP_XOR8>(c, X.f2);
  x_ct<DPP_XOR7>(c, X.f3);
  x_ct<DPP_XOR2>(c, X.f4);
  x_ct<DPP_XOR1>(c, X.f5);  // <= 3330
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int u = 0; u < 4; u++) w_ct(c[8 * h + u], c[8 * h + 4 + u], X.fz4[h]);
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int u = 0; u < 2; u++) w_ct(c[4 * g + u], c[4 * g + 2 + u], X.fz2[g]);
}
// in: |c| <= 1665; out: canonical scale * INTT(c) (fin, gin as w_ntt_inv)
__device__ __forceinline__ void x_ntt_inv(const XLane& X, float (&c)[16], float fin, float gin) {
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int u = 0; u < 2; u++) w_gs(c[4 * g + u], c[4 * g + 2 + u], X.iz2[g]);  // sums 3330
#pragma unroll
  for (int h = 0; h < 2; h++) {
#pragma unroll
    for (int u = 0; u < 4; u++) w_gs(c[8 * h + u], c[8 * h + 4 + u], X.iz4[h]);  // u < 2: 6660
    c[8 * h] = fred(c[8 * h]);
    c[8 * h + 1] = fred(c[8 * h + 1]);
  }
  x_gs<DPP_XOR1>(c, X.i5);  // <= 3330 in, 1665 out
  x_gs<DPP_XOR2>(c, X.i4);
  x_gs<DPP_XOR7>(c, X.i3);
  x_gs<DPP_XOR8>(c, X.i2);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const float s = c[k] + c[k + 8], d = c[k + 8] - c[k];
    c[k] = fmulc_canon(s, fin);
    c[k + 8] = fmulc_canon(d, gin);
  }
}

enum XOp { X_NTT_CANON, X_INVNTT_CANON, X_MUL_CANON };
template <int OP, int U, int WAVES = 1>
__global__ void __launch_bounds__(256, WAVES) k_xpoly(long npoly, int16_t* r, const int16_t* a, const int16_t* b) {
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  XLane X;
  x_lane(X, lane & 15);
  const long p0 = ((long)blockIdx.x * 4 + wave) * (4 * U) + (lane >> 4);
  uint4 a0[U], a1[U], b0[U], b1[U];
  long off[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const long pi = p0 + 4 * u;
    off[u] = (pi < npoly ? pi : npoly - 1) * 256 + 8 * X.m;
    a0[u] = w_ld16(a + off[u]);
    a1[u] = w_ld16(a + off[u] + 128);
    if constexpr (OP == X_MUL_CANON) {
      b0[u] = w_ld16(b + off[u]);
      b1[u] = w_ld16(b + off[u] + 128);
    }
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    float c[16];
    w_unpack(a0[u], a1[u], c);
    if constexpr (OP == X_NTT_CANON) {
      w_centre_if(w_any_above(c, 4095.0f), c);
      x_ntt_fwd(X, c);
      w_fred16_canon(c);
    } else if constexpr (OP == X_INVNTT_CANON) {
      w_centre_if(w_any_above(c, 1665.0f), c);
      x_ntt_inv(X, c, W_INV_TOMONT, W_INV_TOMONT_Z1);
    } else {
      float y[16], x[16];
#pragma unroll
      for (int k = 0; k < 16; k++) x[k] = c[k];
      w_centre_if(w_any_above(x, 4095.0f), x);
      x_ntt_fwd(X, x);
      w_fred16(x);
      w_unpack(b0[u], b1[u], y);
      w_centre_if(w_any_above(y, 4095.0f), y);
      x_ntt_fwd(X, y);  // <= 6660
      WTwFwd T;         // w_basemul reads only z2
      T.z2[0] = X.fz2[0]; T.z2[1] = X.fz2[1]; T.z2[2] = X.fz2[2]; T.z2[3] = X.fz2[3];
      w_basemul(c, x, y, T);
      w_fred16(c);
      x_ntt_inv(X, c, W_INV_PLAIN, W_INV_PLAIN_Z1);
    }
    uint4 lo, hi;
    w_pack(c, lo, hi);
    if (p0 + 4 * u < npoly) {
      w_st16(r + off[u], lo);
      w_st16(r + off[u] + 128, hi);
    }
  }
}

template <class F>
float timeit(F&& launch, hipEvent_t a, hipEvent_t b) {
  float t[5];
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(a);
    for (int i = 0; i < 20; i++) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    t[rep] = ms / 20;
  }
  for (int i = 0; i < 5; i++) for (int j = i + 1; j < 5; j++) if (t[j] < t[i]) { float x = t[i]; t[i] = t[j]; t[j] = x; }
  return t[2];
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 1 << 20;
  const unsigned grid = (unsigned)((n + 15) / 16), grid2 = (unsigned)((n + 31) / 32);
  int16_t *da, *db, *dr;
  CK(hipMalloc(&da, n * 512)); CK(hipMalloc(&db, n * 512)); CK(hipMalloc(&dr, n * 512));
  std::vector<int16_t> ha(n * 256), hb(n * 256), hr(n * 256);
  { uint32_t x = 1; for (auto& v : ha) { x = x * 1664525u + 1013904223u; v = (int16_t)((x >> 8) % 3329); } for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (int16_t)((x >> 8) % 3329); } }
  for (int j = 0; j < 256; j++) { ha[256 + j] = 3328; hb[256 + j] = 3328; ha[512 + j] = 0; ha[768 + j] = (int16_t)(j & 1 ? 6657 : -6657); hb[768 + j] = (int16_t)(4095 - 37 * j); }
  CK(hipMemcpy(da, ha.data(), n * 512, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), n * 512, hipMemcpyHostToDevice));

  const long check[] = {0, 1, 2, 3, 4, 5, 17, n / 2 + 3, n - 2, n - 1};
  int bad = 0;
  auto verify = [&](const char* name, int op) {
    CK(hipMemcpy(hr.data(), dr, n * 512, hipMemcpyDeviceToHost));
    int wrong = 0;
    for (long pi : check) {
      if (pi < 0 || pi >= n) continue;
      long x[256], y[256], w[256];
      for (int j = 0; j < 256; j++) { x[j] = md(ha[pi * 256 + j]); y[j] = md(hb[pi * 256 + j]); }
      if (op == 0) { m_ntt(x); for (int j = 0; j < 256; j++) w[j] = x[j]; }
      else if (op == 1) { m_invntt_tomont(x); for (int j = 0; j < 256; j++) w[j] = x[j]; }
      else { m_ntt(x); m_ntt(y); m_basemul(w, x, y); m_invntt_tomont(w); }
      for (int j = 0; j < 256; j++) {
        const int got = hr[pi * 256 + j];
        if (got != w[j] && wrong++ < 4) printf("  %s: poly %ld coeff %d: got %d want %ld\n", name, pi, j, got, w[j]);
      }
    }
    printf("%-28s %s\n", name, wrong ? "MISMATCH" : "ok (canonical, equal to the integer model)");
    bad += wrong;
    return 0;
  };
  k_xpoly<X_NTT_CANON, 1><<<grid, 256>>>(n, dr, da, nullptr); verify("dpp ntt_canon", 0);
  k_xpoly<X_INVNTT_CANON, 1><<<grid, 256>>>(n, dr, da, nullptr); verify("dpp invntt_canon", 1);
  k_xpoly<X_MUL_CANON, 1><<<grid, 256>>>(n, dr, da, db); verify("dpp mul_fused_canon", 2);
  k_xpoly<X_NTT_CANON, 2><<<grid2, 256>>>(n, dr, da, nullptr); verify("dpp ntt_canon U=2", 0);
  k_xpoly<X_INVNTT_CANON, 2><<<grid2, 256>>>(n, dr, da, nullptr); verify("dpp invntt_canon U=2", 1);
  k_xpoly<X_MUL_CANON, 2><<<grid2, 256>>>(n, dr, da, db); verify("dpp mul_fused_canon U=2", 2);
  CK(hipDeviceSynchronize());
  if (bad) { printf("FAILED: %d wrong coefficients\n", bad); return 2; }

  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto row = [&](const char* name, float ms, double bytes) {
    printf("%-40s %.3f ms  %.2f G/s  %.2f TB/s\n", name, ms, n / ms * 1e-6, n * bytes / ms * 1e-9);
  };
  for (int round = 0; round < 2; round++) {  // twice: the second pass shows what the order of the rows does
    row("LDS  ntt_canon U=2", timeit([&] { k_wpoly<W_NTT_CANON, 2><<<grid2, 256>>>(n, dr, da, nullptr); }, a, b), 1024);
    row("DPP  ntt_canon U=1", timeit([&] { k_xpoly<X_NTT_CANON, 1><<<grid, 256>>>(n, dr, da, nullptr); }, a, b), 1024);
    row("DPP  ntt_canon U=2", timeit([&] { k_xpoly<X_NTT_CANON, 2><<<grid2, 256>>>(n, dr, da, nullptr); }, a, b), 1024);
    row("LDS  invntt_canon U=2", timeit([&] { k_wpoly<W_INVNTT_CANON, 2><<<grid2, 256>>>(n, dr, da, nullptr); }, a, b), 1024);
    row("DPP  invntt_canon U=1", timeit([&] { k_xpoly<X_INVNTT_CANON, 1><<<grid, 256>>>(n, dr, da, nullptr); }, a, b), 1024);
    row("DPP  invntt_canon U=2", timeit([&] { k_xpoly<X_INVNTT_CANON, 2><<<grid2, 256>>>(n, dr, da, nullptr); }, a, b), 1024);
    row("LDS  mul_fused_canon U=1 5 waves", timeit([&] { k_wpoly<W_MUL_CANON, 1, 5><<<grid, 256>>>(n, dr, da, db); }, a, b), 1536);
    row("DPP  mul_fused_canon U=1", timeit([&] { k_xpoly<X_MUL_CANON, 1><<<grid, 256>>>(n, dr, da, db); }, a, b), 1536);
    row("DPP  mul_fused_canon U=1 5 waves", timeit([&] { k_xpoly<X_MUL_CANON, 1, 5><<<grid, 256>>>(n, dr, da, db); }, a, b), 1536);
    row("DPP  mul_fused_canon U=2", timeit([&] { k_xpoly<X_MUL_CANON, 2><<<grid2, 256>>>(n, dr, da, db); }, a, b), 1536);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
