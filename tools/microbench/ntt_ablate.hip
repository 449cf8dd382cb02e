// ntt_ablate.hip -- where does the standalone forward NTT (k_poly_ntt, BASELINE.json
// config 2) spend its time?  The product loop with pieces removed, interleaved in
// one process on the same 2^20 polynomials:
//   full        the product transform
//   copy        load, int16->fp32, fp32->int16, store (the access pattern's own roof)
//   no-lds      butterflies only, the three LDS transposes left out (wrong results)
//   lds-only    the transposes only, no butterflies
// and the grid size (workgroups per CU) as a second axis.
#include "../../kyber_amd/csrc/arith_kernels.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace ky;

template <bool LDS, bool BF>
__device__ __forceinline__ void ntt_variant(float* p, const float* Z, int t, float (&c)[8]) {
  if constexpr (LDS) { KY_WSYNC(); f_put_contig(p, t, c); KY_WSYNC(); f_get_stride(p, t, c); }
  if constexpr (BF) {
#pragma unroll
    for (int m = 0; m < 4; m++) f_ct(c[m], c[m + 4], ZETAF[1]);
    f_ct(c[0], c[2], ZETAF[2]); f_ct(c[1], c[3], ZETAF[2]);
    f_ct(c[4], c[6], ZETAF[3]); f_ct(c[5], c[7], ZETAF[3]);
    f_ct(c[0], c[1], ZETAF[4]); f_ct(c[2], c[3], ZETAF[5]);
    f_ct(c[4], c[5], ZETAF[6]); f_ct(c[6], c[7], ZETAF[7]);
#pragma unroll
    for (int m = 0; m < 8; m++) c[m] = fred(c[m]);
  }
  if constexpr (LDS) { KY_WSYNC(); f_put_stride(p, t, c); KY_WSYNC(); f_get_middle(p, t, c); }
  if constexpr (BF) {
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
  if constexpr (LDS) { KY_WSYNC(); f_put_middle(p, t, c); KY_WSYNC(); f_get_contig(p, t, c); }
  if constexpr (BF) {
    float z0 = Z[64 + 2 * t], z1 = Z[65 + 2 * t];
    f_ct(c[0], c[2], z0); f_ct(c[1], c[3], z0);
    f_ct(c[4], c[6], z1); f_ct(c[5], c[7], z1);
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = fred(c[k]);
  }
}

template <bool LDS, bool BF, int U>
__global__ void __launch_bounds__(256) k_var(long npoly, int16_t* __restrict__ r) {
  __shared__ __attribute__((aligned(16))) float wk[8][256];
  __shared__ float ztab[128];
  const Grp w = grp();
  f_twiddles_to_lds(ztab);
  const long stride = (long)gridDim.x * 8;
  const long last = npoly - 1;
  for (long pi = (long)blockIdx.x * 8 + w.g; pi <= last; pi += U * stride) {
    uint4 in[U];
#pragma unroll
    for (int u = 0; u < U; u++) in[u] = ld_poly16(r, min(pi + u * stride, last), w.t);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long at = pi + u * stride;
      float f[8];
      s16x8_to_f(in[u], f);
      ntt_variant<LDS, BF>(wk[w.g], ztab, w.t, f);
      if (at <= last) *reinterpret_cast<uint4*>(r + at * 256 + 8 * w.t) = f_to_s16x8(f);
    }
  }
}

template <bool LDS, bool BF, int U>
float run(long n, int16_t* d, int wg_per_cu, hipEvent_t a, hipEvent_t b) {
  // 20 launches back to back (as bench.py times them), median of 5 such runs, per launch
  float t[5];
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) k_var<LDS, BF, U><<<256 * wg_per_cu, 256>>>(n, d);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    t[rep] = ms / 20;
  }
  for (int i = 0; i < 5; i++) for (int j = i + 1; j < 5; j++) if (t[j] < t[i]) { float x = t[i]; t[i] = t[j]; t[j] = x; }
  return t[2];
}

int main() {
  const long n = 1 << 20;
  int16_t* d; CK(hipMalloc(&d, n * 512));
  { std::vector<int16_t> h(n * 256); uint32_t x = 1; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (int16_t)((x >> 8) % 3329); } CK(hipMemcpy(d, h.data(), n * 512, hipMemcpyHostToDevice)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int wg : {4, 7, 8, 14, 16, 32}) {
    float full = run<true, true, 2>(n, d, wg, a, b), copy = run<false, false, 2>(n, d, wg, a, b);
    float nolds = run<false, true, 2>(n, d, wg, a, b), ldsonly = run<true, false, 2>(n, d, wg, a, b);
    float full4 = run<true, true, 4>(n, d, wg, a, b), full1 = run<true, true, 1>(n, d, wg, a, b);
    printf("wg/CU %2d: full(U=2) %.3f ms = %.2f G/s | U=1 %.3f U=4 %.3f | copy %.3f ms = %.2f TB/s | no-lds %.3f | lds-only %.3f\n", wg, full,
           n / full * 1e-6, full1, full4, copy, n * 1024.0 / copy * 1e-9, nolds, ldsonly);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
