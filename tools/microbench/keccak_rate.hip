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
  // known-answer: permutation of the all-zero state, first lane = 0xF1258F7940E1DDE7
  uint2* d; CK(hipMalloc(&d, 256 * 25 * 8)); CK(hipMemset(d, 0, 256 * 25 * 8));
  k_perm<2><<<1, 256>>>(d, 256, 1);
  uint2 h; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
  printf("KAT lane0 = %08X%08X (expect F1258F7940E1DDE7)\n", h.y, h.x);
  return 0;
}
