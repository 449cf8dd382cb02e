// wpoly_rate.hip -- the wide-layout polynomial kernels (kyber_amd/csrc/wpoly.h) on 2^20 polynomials:
// every kernel checked mod q against a plain integer model of the reference's loops (ref/ntt.c:80-146)
// on the first and last polynomials, then timed (20 launches back to back, median of 5), next to the
// copy of the same walk (load, int16 -> fp32 -> int16, store) and the round-2 kernels.
#include "../../kyber_amd/csrc/arith_kernels.h"
#include "../../kyber_amd/csrc/wpoly.h"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using namespace ky;

// ---- integer model (values mod q; zetas in Montgomery form as the reference holds them)
static const int ZM[128] = KY_ZETAS_INIT;
static long md(long x) { x %= 3329; return x < 0 ? x + 3329 : x; }
static long fq(long a, long b) { return md(md(a * b) * 169); }  // fqmul: a b 2^-16
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

__global__ void __launch_bounds__(256) k_wcopy(long npoly, int16_t* r, const int16_t* a) {
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const long pi = (long)blockIdx.x * 16 + 4 * wave + (lane >> 4);
  const long pc = pi < npoly ? pi : npoly - 1;
  const long off = pc * 256 + 8 * (lane & 15);
  const uint4 a0 = w_ld16(a + off), a1 = w_ld16(a + off + 128);
  float c[16];
  w_unpack(a0, a1, c);
  uint4 lo, hi;
  w_pack(c, lo, hi);
  if (pi < npoly) { w_st16(r + off, lo); w_st16(r + off + 128, hi); }
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
  // edge polynomials: all q-1, all 0, and wide inputs (the kernels centre what is beyond their entry bound)
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
      else if (op == 2) { m_basemul(w, x, y); }
      else { m_ntt(x); m_ntt(y); m_basemul(w, x, y); m_invntt_tomont(w); }
      for (int j = 0; j < 256; j++) {
        const int got = hr[pi * 256 + j];
        const bool ok = op == 4 ? md(got) == w[j] : got == w[j];
        if (!ok && wrong++ < 4) printf("  %s: poly %ld coeff %d: got %d want %ld\n", name, pi, j, got, w[j]);
      }
    }
    printf("%-28s %s\n", name, wrong ? "MISMATCH" : "ok (canonical, equal to the integer model)");
    bad += wrong;
    return 0;
  };
  k_wpoly<W_NTT_CANON, 1><<<grid, 256>>>(n, dr, da, nullptr); verify("ntt_canon", 0);
  k_wpoly<W_INVNTT_CANON, 1><<<grid, 256>>>(n, dr, da, nullptr); verify("invntt_canon", 1);
  k_wpoly<W_BASEMUL_CANON, 1><<<grid, 256>>>(n, dr, da, db); verify("basemul_canon", 2);
  k_wpoly<W_MUL_CANON, 1><<<grid, 256>>>(n, dr, da, db); verify("mul_fused_canon", 3);
  k_wpoly<W_NTT_CANON, 2><<<grid2, 256>>>(n, dr, da, nullptr); verify("ntt_canon U=2", 0);
  k_wpoly<W_INVNTT_CANON, 2><<<grid2, 256>>>(n, dr, da, nullptr); verify("invntt_canon U=2", 1);
  k_wpoly<W_BASEMUL_CANON, 2><<<grid2, 256>>>(n, dr, da, db); verify("basemul_canon U=2", 2);
  k_wpoly<W_MUL_CANON, 2><<<grid2, 256>>>(n, dr, da, db); verify("mul_fused_canon U=2", 3);
  {  // the Barrett-centred forward transform: same residues, representative in [-1664, 1664]
    k_wpoly<W_NTT, 2><<<grid2, 256>>>(n, dr, da, nullptr);
    CK(hipMemcpy(hr.data(), dr, n * 512, hipMemcpyDeviceToHost));
    int wrong = 0;
    for (long pi : check) {
      long x[256];
      for (int j = 0; j < 256; j++) x[j] = md(ha[pi * 256 + j]);
      m_ntt(x);
      for (int j = 0; j < 256; j++) { const int got = hr[pi * 256 + j]; if (md(got) != x[j] || got > 1664 || got < -1664) wrong++; }
    }
    printf("%-28s %s\n", "ntt (centred)", wrong ? "MISMATCH" : "ok");
    bad += wrong;
  }
  CK(hipDeviceSynchronize());
  if (bad) { printf("FAILED: %d wrong coefficients\n", bad); return 2; }

  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto row = [&](const char* name, float ms, double bytes) {
    printf("%-34s %.3f ms  %.2f G/s  %.2f TB/s\n", name, ms, n / ms * 1e-6, n * bytes / ms * 1e-9);
  };
  row("copy (the walk's own roof)", timeit([&] { k_wcopy<<<grid, 256>>>(n, dr, da); }, a, b), 1024);
  row("copy in place", timeit([&] { k_wcopy<<<grid, 256>>>(n, dr, dr); }, a, b), 1024);
#define BOTH(name, OP, A, B, bytes)                                                                          \
  row(name " U=1", timeit([&] { k_wpoly<OP, 1><<<grid, 256>>>(n, dr, A, B); }, a, b), bytes);                  \
  row(name " U=2", timeit([&] { k_wpoly<OP, 2><<<grid2, 256>>>(n, dr, A, B); }, a, b), bytes)
  row("ntt_canon U=1 8 waves", timeit([&] { k_wpoly<W_NTT_CANON, 1, 8><<<grid, 256>>>(n, dr, dr, nullptr); }, a, b), 1024);
  row("ntt_canon U=2 6 waves", timeit([&] { k_wpoly<W_NTT_CANON, 2, 6><<<grid2, 256>>>(n, dr, dr, nullptr); }, a, b), 1024);
  row("ntt_canon U=3", timeit([&] { k_wpoly<W_NTT_CANON, 3><<<(unsigned)((n + 47) / 48), 256>>>(n, dr, dr, nullptr); }, a, b), 1024);
  row("ntt_canon U=4", timeit([&] { k_wpoly<W_NTT_CANON, 4><<<(unsigned)((n + 63) / 64), 256>>>(n, dr, dr, nullptr); }, a, b), 1024);
  row("invntt_canon U=1 8 waves", timeit([&] { k_wpoly<W_INVNTT_CANON, 1, 8><<<grid, 256>>>(n, dr, dr, nullptr); }, a, b), 1024);
  row("invntt_canon U=4", timeit([&] { k_wpoly<W_INVNTT_CANON, 4><<<(unsigned)((n + 63) / 64), 256>>>(n, dr, dr, nullptr); }, a, b), 1024);
  row("mul_fused_canon U=1 5 waves", timeit([&] { k_wpoly<W_MUL_CANON, 1, 5><<<grid, 256>>>(n, dr, da, db); }, a, b), 1536);
  row("mul_fused_canon U=2 4 waves", timeit([&] { k_wpoly<W_MUL_CANON, 2, 4><<<grid2, 256>>>(n, dr, da, db); }, a, b), 1536);
  BOTH("ntt (centred) in place", W_NTT, dr, nullptr, 1024);
  BOTH("ntt_canon in place", W_NTT_CANON, dr, nullptr, 1024);
  BOTH("invntt_canon in place", W_INVNTT_CANON, dr, nullptr, 1024);
  BOTH("basemul_canon", W_BASEMUL_CANON, da, db, 1536);
  BOTH("mul_fused_canon", W_MUL_CANON, da, db, 1536);
  const unsigned g2 = 4096;
  // (round 2's forward kernel, 32 lanes per polynomial in a grid-stride loop, read 4.7-4.8 G/s here: profiles/r03_microbench_wpoly_rate_v1.txt)
  row("round 2: k_poly_invntt (int16)", timeit([&] { k_poly_invntt<<<g2, 256>>>(n, dr); }, a, b), 1024);
  row("round 2: k_poly_basemul (int16)", timeit([&] { k_poly_basemul<<<(unsigned)((n + 7) / 8), 256>>>(n, dr, da, db); }, a, b), 1536);
  row("round 2: k_poly_mul_fused (int16)", timeit([&] { k_poly_mul_fused<<<g2, 256>>>(n, dr, da, db); }, a, b), 1536);
  CK(hipDeviceSynchronize());
  return 0;
}
