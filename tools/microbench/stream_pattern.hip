// stream_pattern.hip -- what does the walk over a dense array of int16[256] polynomials cost?
// (BASELINE.json config 2: the transform kernels are bound by what their access pattern gets out
// of HBM.)  One 32-lane group per polynomial, 16 bytes per lane, as in arith_kernels.h; varied:
//   grid      one-shot (one workgroup per 8 polynomials, no loop) | grid-stride loop (wg/CU given) |
//             block-contiguous loop (each workgroup walks its own contiguous run)
//   target    in place | out of place
//   cache     plain | non-temporal loads and stores
//   work      copy (int16 -> fp32 -> int16) | the forward transform
// 2^20 polynomials, 20 launches back to back, median of 5.
#include "../../kyber_amd/csrc/arith_kernels.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace ky;

template <bool NT>
__device__ __forceinline__ uint4 ldp(const int16_t* base, long pi, int t) {
  const uint4* p = reinterpret_cast<const uint4*>(base + pi * 256 + 8 * t);
  if constexpr (NT) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *p;
  }
}
template <bool NT>
__device__ __forceinline__ void stp(int16_t* base, long pi, int t, uint4 v) {
  uint4* p = reinterpret_cast<uint4*>(base + pi * 256 + 8 * t);
  if constexpr (NT) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    v4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<v4*>(p));
  } else {
    *p = v;
  }
}

template <bool WORK>
__device__ __forceinline__ uint4 process(uint4 in, float* wk, const float* ztab, int t) {
  float f[8];
  s16x8_to_f(in, f);
  if constexpr (WORK) f_ntt_fwd(wk, ztab, t, f);
  return f_to_s16x8(f);
}

// GRID: 0 one-shot, 1 grid-stride, 2 block-contiguous
template <int GRID, bool NT, bool WORK, int U>
__global__ void __launch_bounds__(256) k_walk(long npoly, int16_t* dst, const int16_t* src) {
  __shared__ __attribute__((aligned(16))) float wk[8][256];
  __shared__ float ztab[128];
  const Grp w = grp();
  if constexpr (GRID == 0) {
    const long pi = (long)blockIdx.x * 8 + w.g;
    const uint4 in = ldp<NT>(src, min(pi, npoly - 1), w.t);  // requested before the twiddle table is set up
    if constexpr (WORK) f_twiddles_to_lds(ztab);
    const uint4 out = process<WORK>(in, wk[w.g], ztab, w.t);
    if (pi < npoly) stp<NT>(dst, pi, w.t, out);
  } else {
    if constexpr (WORK) f_twiddles_to_lds(ztab);
    const long last = npoly - 1;
    long pi, end, step;
    if constexpr (GRID == 1) {
      step = (long)gridDim.x * 8;
      pi = (long)blockIdx.x * 8 + w.g;
      end = npoly;
    } else {
      const long per = (npoly + gridDim.x - 1) / gridDim.x;  // polynomials per workgroup, a multiple of 8 for the sizes used
      pi = (long)blockIdx.x * per + w.g;
      end = min(npoly, (long)(blockIdx.x + 1) * per);
      step = 8;
    }
    for (; pi < end; pi += U * step) {
      uint4 in[U];
#pragma unroll
      for (int u = 0; u < U; u++) in[u] = ldp<NT>(src, min(pi + u * step, last), w.t);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long at = pi + u * step;
        const uint4 out = process<WORK>(in[u], wk[w.g], ztab, w.t);
        if (at < end) stp<NT>(dst, at, w.t, out);
      }
    }
  }
}

template <int GRID, bool NT, bool WORK, int U>
float run(long n, int16_t* dst, const int16_t* src, int wg_per_cu, hipEvent_t a, hipEvent_t b) {
  float t[5];
  const unsigned grid = GRID == 0 ? (unsigned)((n + 7) / 8) : 256u * wg_per_cu;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) k_walk<GRID, NT, WORK, U><<<grid, 256>>>(n, dst, src);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    t[rep] = ms / 20;
  }
  for (int i = 0; i < 5; i++) for (int j = i + 1; j < 5; j++) if (t[j] < t[i]) { float x = t[i]; t[i] = t[j]; t[j] = x; }
  return t[2];
}

#define ROW(label, GRID, NT, WORK, U, wg)                                                               \
  do {                                                                                                  \
    float ip = run<GRID, NT, WORK, U>(n, d, d, wg, a, b), op = run<GRID, NT, WORK, U>(n, d2, d, wg, a, b); \
    printf("%-44s in place %.3f ms = %.2f TB/s = %.2f G/s | out of place %.3f ms = %.2f TB/s = %.2f G/s\n", label, ip,   \
           n * 1024.0 / ip * 1e-9, n / ip * 1e-6, op, n * 1024.0 / op * 1e-9, n / op * 1e-6);           \
  } while (0)

int main() {
  const long n = 1 << 20;
  int16_t *d, *d2;
  CK(hipMalloc(&d, n * 512));
  CK(hipMalloc(&d2, n * 512));
  { std::vector<int16_t> h(n * 256); uint32_t x = 1; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (int16_t)((x >> 8) % 3329); } CK(hipMemcpy(d, h.data(), n * 512, hipMemcpyHostToDevice)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  ROW("copy one-shot", 0, false, false, 1, 0);
  ROW("copy one-shot nt", 0, true, false, 1, 0);
  ROW("copy stride 8wg/CU U1", 1, false, false, 1, 8);
  ROW("copy stride 8wg/CU U1 nt", 1, true, false, 1, 8);
  ROW("copy stride 8wg/CU U2", 1, false, false, 2, 8);
  ROW("copy stride 8wg/CU U2 nt", 1, true, false, 2, 8);
  ROW("copy stride 8wg/CU U4 nt", 1, true, false, 4, 8);
  ROW("copy stride 16wg/CU U1", 1, false, false, 1, 16);
  ROW("copy stride 16wg/CU U2 nt", 1, true, false, 2, 16);
  ROW("copy contig 8wg/CU U1", 2, false, false, 1, 8);
  ROW("copy contig 8wg/CU U2", 2, false, false, 2, 8);
  ROW("copy contig 8wg/CU U2 nt", 2, true, false, 2, 8);
  ROW("copy contig 16wg/CU U2 nt", 2, true, false, 2, 16);
  ROW("ntt one-shot", 0, false, true, 1, 0);
  ROW("ntt one-shot nt", 0, true, true, 1, 0);
  ROW("ntt stride 8wg/CU U1", 1, false, true, 1, 8);
  ROW("ntt stride 8wg/CU U1 nt", 1, true, true, 1, 8);
  ROW("ntt stride 8wg/CU U2", 1, false, true, 2, 8);
  ROW("ntt stride 8wg/CU U2 nt", 1, true, true, 2, 8);
  ROW("ntt stride 16wg/CU U1", 1, false, true, 1, 16);
  ROW("ntt stride 16wg/CU U1 nt", 1, true, true, 1, 16);
  ROW("ntt stride 16wg/CU U2 nt", 1, true, true, 2, 16);
  ROW("ntt contig 8wg/CU U2 nt", 2, true, true, 2, 8);
  ROW("ntt contig 16wg/CU U2 nt", 2, true, true, 2, 16);
  CK(hipDeviceSynchronize());
  return 0;
}
