// vgpr_bank.hip -- does the placement of a 3-source VALU instruction's operands in
// the VGPR file matter on gfx950?  Eight independent v_bitop3_b32 / v_alignbit_b32
// chains with explicitly named registers: sources spread over vN mod 4 = {0,1,2,3}
// versus all sources in registers of the same vN mod 4 class.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// d = op(d, a, b): chains d in v[16..23]; operands a,b in named registers
#define CH(op, d, a, b, extra) op " v" #d ", v" #d ", v" #a ", v" #b extra "\n"
// spread: d, a, b in three different mod-4 classes for every chain
#define BODY_SPREAD(op, extra) \
  CH(op, 16, 25, 34, extra) CH(op, 17, 26, 35, extra) CH(op, 18, 27, 32, extra) CH(op, 19, 24, 33, extra) \
  CH(op, 20, 29, 38, extra) CH(op, 21, 30, 39, extra) CH(op, 22, 31, 36, extra) CH(op, 23, 28, 37, extra)
// same: d, a, b all in the same mod-4 class
#define BODY_SAME(op, extra) \
  CH(op, 16, 24, 32, extra) CH(op, 17, 25, 33, extra) CH(op, 18, 26, 34, extra) CH(op, 19, 27, 35, extra) \
  CH(op, 20, 28, 36, extra) CH(op, 21, 29, 37, extra) CH(op, 22, 30, 38, extra) CH(op, 23, 31, 39, extra)
// two: a and b in the same class, d in another
#define BODY_TWO(op, extra) \
  CH(op, 16, 25, 33, extra) CH(op, 17, 26, 34, extra) CH(op, 18, 27, 35, extra) CH(op, 19, 24, 32, extra) \
  CH(op, 20, 29, 37, extra) CH(op, 21, 30, 38, extra) CH(op, 22, 31, 39, extra) CH(op, 23, 28, 36, extra)

#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39"
#define KERNEL(NAME, BODY)                                                              \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters) {               \
    asm volatile("v_mov_b32 v24, %0\nv_mov_b32 v25, %0\nv_mov_b32 v26, %0\nv_mov_b32 v27, %0\nv_mov_b32 v28, %0\nv_mov_b32 v29, %0\nv_mov_b32 v30, %0\nv_mov_b32 v31, %0\n" \
                 "v_mov_b32 v32, %1\nv_mov_b32 v33, %1\nv_mov_b32 v34, %1\nv_mov_b32 v35, %1\nv_mov_b32 v36, %1\nv_mov_b32 v37, %1\nv_mov_b32 v38, %1\nv_mov_b32 v39, %1\n" \
                 "v_mov_b32 v16, %0\nv_mov_b32 v17, %1\nv_mov_b32 v18, %0\nv_mov_b32 v19, %1\nv_mov_b32 v20, %0\nv_mov_b32 v21, %1\nv_mov_b32 v22, %0\nv_mov_b32 v23, %1\n" \
                 ::"v"(threadIdx.x * 2654435761u), "v"(threadIdx.x ^ 0x5bd1e995u) : CLOB);            \
    for (int i = 0; i < iters; i++) asm volatile(BODY BODY BODY BODY ::: CLOB);          \
    uint32_t r;                                                                          \
    asm volatile("v_xor_b32 %0, v16, v17\nv_xor_b32 %0, %0, v18\nv_xor_b32 %0, %0, v19\nv_xor_b32 %0, %0, v20\nv_xor_b32 %0, %0, v21\nv_xor_b32 %0, %0, v22\nv_xor_b32 %0, %0, v23\n" : "=v"(r)::CLOB); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                      \
  }

KERNEL(k_bitop_spread, BODY_SPREAD("v_bitop3_b32", " bitop3:0x96"))
KERNEL(k_bitop_two, BODY_TWO("v_bitop3_b32", " bitop3:0x96"))
KERNEL(k_bitop_same, BODY_SAME("v_bitop3_b32", " bitop3:0x96"))
KERNEL(k_align_spread, BODY_SPREAD("v_alignbit_b32", ""))
KERNEL(k_align_same, BODY_SAME("v_alignbit_b32", ""))

template <class K>
float run(K k, uint32_t* d, int wg_per_cu, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < 4; r++) {
    hipEventRecord(a);
    k<<<256 * wg_per_cu, 256>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  uint32_t* d; CK(hipMalloc(&d, 256 * 16 * 256 * 4));
  const int iters = 4000;  // x 32 instructions
  for (int wg : {2, 4, 8}) {
    const double ops = 256.0 * wg * 256 * iters * 32;
    float a = run(k_bitop_spread, d, wg, iters), b = run(k_bitop_two, d, wg, iters), c = run(k_bitop_same, d, wg, iters);
    float e = run(k_align_spread, d, wg, iters), f = run(k_align_same, d, wg, iters);
    printf("%d waves/SIMD: v_bitop3 spread %.2f T/s | two sources same class %.2f | all same class %.2f || v_alignbit spread %.2f | same class %.2f  (lane-op/s)\n",
           wg, ops / a * 1e-9, ops / b * 1e-9, ops / c * 1e-9, ops / e * 1e-9, ops / f * 1e-9);
  }
  return 0;
}
