// fusion_bound.hip -- what would a fused "sample the matrix into LDS, consume it in
// the same workgroup" kernel run at?  
// A is 512 bytes per entry and a sponge-layout wavefront holds 64 entries, so a
// workgroup of 256 sponges needs 28 instances x 9 entries x 512 B = 129 KB of LDS for
// the tile alone: ONE such workgroup per CU, one wavefront per SIMD (and the phase-locked
// 512-lane workgroups of keccak.h would need 256 KB: impossible).  This program runs
// the two product kernels at one workgroup per CU (dynamic LDS padding, nothing else
// changed; for the sampler that is one 512-lane workgroup, i.e. still twice the residency
// the fused tile allows) next to their normal residency, which prices the fusion's
// occupancy cost without writing the fused kernel; the traffic it would save is
// priced by matrix_ablate (drain) and by the shared-matrix run of bench.py.
#include "../../kyber_amd/csrc/kem_kernels.h"
#include "../../kyber_amd/csrc/arith_kernels.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using P = ky::Par<3>;

int main() {
  const int n = 1 << 17;
  uint8_t *seeds, *pk, *small, *prf, *ct; int16_t* mat;
  CK(hipMalloc(&seeds, n * 32)); CK(hipMalloc(&mat, (size_t)n * 9 * 512)); CK(hipMalloc(&pk, (size_t)n * P::PKB));
  CK(hipMalloc(&small, (size_t)n * 128)); CK(hipMalloc(&prf, (size_t)n * 7 * 128)); CK(hipMalloc(&ct, (size_t)n * P::CTB));
  { std::vector<uint32_t> h(n * 8); for (int i = 0; i < n * 8; i++) h[i] = i * 2654435761u; CK(hipMemcpy(seeds, h.data(), n * 32, hipMemcpyHostToDevice)); }
  CK(hipMemset(pk, 0x11, (size_t)n * P::PKB)); CK(hipMemset(small, 0x22, (size_t)n * 128)); CK(hipMemset(prf, 0x33, (size_t)n * 7 * 128));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto k_s = ky::k_sample_matrix<P, ky::SPONGE_WG, 0>;
  auto k_e = ky::k_enc_arith<P, false>;
  const int jobs = n * 9;
  for (int r = 0; r < 3; r++) {
    float t[4];
    for (int v = 0; v < 4; v++) {
      const size_t pad = (v & 1) ? ((v < 2) ? 40 * 1024 : 130 * 1024) : 0;  // forces one workgroup per CU
      // the attribute is raised for the padded launch only: left raised, it slows the unpadded launches of the same kernel too
      CK(hipFuncSetAttribute(v < 2 ? reinterpret_cast<const void*>(k_s) : reinterpret_cast<const void*>(k_e), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
      CK(hipEventRecord(a));
      if (v < 2) k_s<<<(jobs + ky::SPONGE_WG - 1) / ky::SPONGE_WG, ky::SPONGE_WG, pad>>>(jobs, seeds, 32, 1, mat, nullptr, 0u);
      else k_e<<<(n + 7) / 8, 256, pad>>>(n, pk, P::PKB, small, mat, (size_t)9 * 256, prf, ct);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&t[v], a, b));
    }
    printf("round %d (n = %d): k_sample_matrix %.3f ms, at one workgroup per CU %.3f ms (x%.2f) | k_enc_arith %.3f ms, at one workgroup per CU %.3f ms (x%.2f)\n",
           r, n, t[0], t[1], t[1] / t[0], t[2], t[3], t[3] / t[2]);
  }
  // Do the two kinds of kernel overlap to advantage?  The sampler on one stream and the encryption
  // arithmetic (twice: about the same duration) on another, against the same three launches in sequence.
  {
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_s), hipFuncAttributeMaxDynamicSharedMemorySize, 0));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_e), hipFuncAttributeMaxDynamicSharedMemorySize, 0));
    for (int r = 0; r < 4; r++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1));
      k_s<<<(jobs + ky::SPONGE_WG - 1) / ky::SPONGE_WG, ky::SPONGE_WG, 0, s1>>>(jobs, seeds, 32, 1, mat, nullptr, 0u);
      k_e<<<(n + 7) / 8, 256, 0, s1>>>(n, pk, P::PKB, small, mat, (size_t)9 * 256, prf, ct);
      k_e<<<(n + 7) / 8, 256, 0, s1>>>(n, pk, P::PKB, small, mat, (size_t)9 * 256, prf, ct);
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      float seq; CK(hipEventElapsedTime(&seq, e0, e1));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1));
      CK(hipStreamWaitEvent(s2, e0, 0));
      k_s<<<(jobs + ky::SPONGE_WG - 1) / ky::SPONGE_WG, ky::SPONGE_WG, 0, s1>>>(jobs, seeds, 32, 1, mat, nullptr, 0u);
      k_e<<<(n + 7) / 8, 256, 0, s2>>>(n, pk, P::PKB, small, mat, (size_t)9 * 256, prf, ct);
      k_e<<<(n + 7) / 8, 256, 0, s2>>>(n, pk, P::PKB, small, mat, (size_t)9 * 256, prf, ct);
      CK(hipEventRecord(e2, s2));
      CK(hipStreamWaitEvent(s1, e2, 0));
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      float con; CK(hipEventElapsedTime(&con, e0, e1));
      printf("round %d: sampler + 2 x k_enc_arith in sequence %.3f ms, on two streams %.3f ms (x%.2f)\n", r, seq, con, con / seq);
    }
  }
  return 0;
}
