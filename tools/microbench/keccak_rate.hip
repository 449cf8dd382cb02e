// keccak_rate.hip -- microbenchmark: Keccak-f[1600] permutations per second,
// one sponge per work-item, at several register budgets.  Used to validate the
// VALU-rate assumptions in DESIGN.md (v_bitop3_b32 / v_alignbit_b32 full rate).
#include "../../kyber_amd/csrc/keccak.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_perm(uint2* st, int n, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ky::u64p A[25];
#pragma unroll
  for (int j = 0; j < 25; j++) { uint2 v = st[(size_t)j * n + i]; A[j] = ky::u64p{v.x, v.y}; }
  for (int it = 0; it < iters; it++) ky::keccak_f1600(A);
#pragma unroll
  for (int j = 0; j < 25; j++) st[(size_t)j * n + i] = make_uint2(A[j].lo, A[j].hi);
}

// the same loop bracketed by the shader-clock counter and the constant-rate wall clock: their ratio
// is the frequency the CUs actually hold while every SIMD issues Keccak rounds
__global__ void __launch_bounds__(256, 4) k_perm_clk(uint2* st, int n, int iters, unsigned long long* clk) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  ky::u64p A[25];
#pragma unroll
  for (int j = 0; j < 25; j++) { uint2 v = st[(size_t)j * n + i]; A[j] = ky::u64p{v.x, v.y}; }
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) ky::keccak_f1600(A);
  unsigned long long c1 = clock64(), w1 = wall_clock64();
#pragma unroll
  for (int j = 0; j < 25; j++) st[(size_t)j * n + i] = make_uint2(A[j].lo, A[j].hi);
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

// workgroups of 512 / 1024 work-items: the 2 / 4 wavefronts a SIMD holds belong to one workgroup, start
// together and run the same instruction stream, i.e. they stay in the same phase of the round
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_perm_wg(uint2* st, int n, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  ky::u64p A[25];
#pragma unroll
  for (int j = 0; j < 25; j++) { uint2 v = st[(size_t)j * n + i]; A[j] = ky::u64p{v.x, v.y}; }
  for (int it = 0; it < iters; it++) ky::keccak_f1600(A);
#pragma unroll
  for (int j = 0; j < 25; j++) st[(size_t)j * n + i] = make_uint2(A[j].lo, A[j].hi);
}
template <int BLOCK>
int run_wg(uint2* d, int n) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_perm_wg<BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int per_cu = 1; per_cu <= 2; per_cu++) {
    if (BLOCK * per_cu > 2048) break;
    const size_t pad = per_cu == 1 ? 100 * 1024 : 60 * 1024;   // one or two workgroups per CU
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {
      CK(hipEventRecord(a));
      k_perm_wg<BLOCK><<<n / BLOCK, BLOCK, pad>>>(d, n, 64);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    printf("workgroups of %4d, %d per CU (%d wavefronts per SIMD): %.3f G perm/s\n", BLOCK, per_cu, BLOCK / 256 * per_cu,
           (double)n * 64 / (best * 1e-3) * 1e-9);
  }
  return 0;
}

template <int WAVES>
int run(uint2* d, int n, int iters) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k_perm<WAVES><<<n / 256, 256>>>(d, n, 2);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(a));
    k_perm<WAVES><<<n / 256, 256>>>(d, n, iters);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  double perms = (double)n * iters / (best * 1e-3);
  printf("launch_bounds(256,%d): n=%d iters=%d  %.3f ms  %.3f G perm/s  %.2f T lane-op/s (4320 ops/perm)\n",
         WAVES, n, iters, best, perms * 1e-9, perms * 4320 * 1e-12);
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs %d  clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  for (int n : {256 * 256 * 4, 256 * 256 * 8, 256 * 256 * 16}) {
    uint2* d; CK(hipMalloc(&d, (size_t)n * 25 * 8));
    CK(hipMemset(d, 0x5a, (size_t)n * 25 * 8));
    if (run<1>(d, n, 64)) return 1;
    if (run<2>(d, n, 64)) return 1;
    if (run<3>(d, n, 64)) return 1;
    if (run<4>(d, n, 64)) return 1;
    CK(hipFree(d));
  }
  // sustained rate: the same launch back to back for about two seconds (the chip lowers its clock
  // under sustained load; a single 7 ms launch from idle runs on boost)
  {
    const int n = 256 * 256 * 16;
    uint2* d; CK(hipMalloc(&d, (size_t)n * 25 * 8)); CK(hipMemset(d, 0x5a, (size_t)n * 25 * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int phase = 0; phase < 6; phase++) {
      CK(hipEventRecord(a));
      for (int r = 0; r < 50; r++) k_perm<4><<<n / 256, 256>>>(d, n, 64);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("sustained, launches %3d-%3d: %.3f G perm/s\n", phase * 50, phase * 50 + 49, (double)n * 64 * 50 / (ms * 1e-3) * 1e-9);
    }
    CK(hipFree(d));
  }
  // rate against the number of resident wavefronts per SIMD (dynamic LDS padding limits residency;
  // the kernel's registers allow 6): how much do the sponge kernels lose to their 4-6 waves per SIMD?
  {
    const int n = 256 * 256 * 16;
    uint2* d; CK(hipMalloc(&d, (size_t)n * 25 * 8)); CK(hipMemset(d, 0x5a, (size_t)n * 25 * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_perm<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 1; w <= 8; w++) {
      const size_t pad = w == 8 ? 0 : (size_t)(160 * 1024 / w) & ~(size_t)255;
      float best = 1e30f;
      for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(a));
        k_perm<4><<<n / 256, 256, pad>>>(d, n, 64);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      printf("at most %d workgroups (= waves per SIMD) per CU by LDS: %.3f G perm/s\n", w, (double)n * 64 / (best * 1e-3) * 1e-9);
    }
    CK(hipFree(d));
  }
  {
    const int n = 256 * 256 * 12;   // a multiple of every workgroup size below
    uint2* d; CK(hipMalloc(&d, (size_t)n * 25 * 8)); CK(hipMemset(d, 0x5a, (size_t)n * 25 * 8));
    if (run_wg<256>(d, n)) return 1;
    if (run_wg<512>(d, n)) return 1;
    if (run_wg<768>(d, n)) return 1;
    if (run_wg<1024>(d, n)) return 1;
    CK(hipFree(d));
  }
  // shader clock under this load
  {
    const int n = 256 * 256 * 16, nb = n / 256;
    int wall_khz = 0; CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    uint2* d; CK(hipMalloc(&d, (size_t)n * 25 * 8)); CK(hipMemset(d, 0x5a, (size_t)n * 25 * 8));
    unsigned long long* clk; CK(hipMalloc(&clk, (size_t)nb * 16));
    std::vector<unsigned long long> h(2 * (size_t)nb);
    for (int r = 0; r < 40; r++) k_perm_clk<<<nb, 256>>>(d, n, 64, clk);   // reach the sustained state first
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), clk, (size_t)nb * 16, hipMemcpyDeviceToHost));
    double sc = 0, sw = 0;
    for (int b = 0; b < nb; b++) { sc += (double)h[2 * b]; sw += (double)h[2 * b + 1]; }
    printf("wall clock %d kHz; shader-clock ticks / wall-clock ticks over the permutation loop = %.4f -> %.1f MHz\n",
           wall_khz, sc / sw, sc / sw * wall_khz * 1e-3);
    printf("wave-instructions per SIMD cycle at that clock: %.3f (4320 per permutation, 10.1 G perm/s)\n",
           10.1e9 / 64 / 1024 * 4320 / (sc / sw * wall_khz * 1e3));
    CK(hipFree(d)); CK(hipFree(clk));
  }
  // known-answer: permutation of the all-zero state, first lane = 0xF1258F7940E1DDE7
  uint2* d; CK(hipMalloc(&d, 256 * 25 * 8)); CK(hipMemset(d, 0, 256 * 25 * 8));
  k_perm<2><<<1, 256>>>(d, 256, 1);
  uint2 h; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
  printf("KAT lane0 = %08X%08X (expect F1258F7940E1DDE7)\n", h.y, h.x);
  return 0;
}
