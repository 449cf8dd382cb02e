// matrix90_ablate.hip -- where does k90_sample_matrix (AES-256-CTR matrix expansion of the 90s sets)
// spend its time?  The product kernel and its ablations (1: no drain; 2: keystream only; 3: parse and
// drain of a made-up keystream, no AES; 4: parse of it alone), interleaved in one process on the same seeds, at the product's
// batch (2^19 x 9 entries).
#include "../../kyber_amd/csrc/kem_kernels.h"
#include "../../kyber_amd/csrc/kem_kernels_90s.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using P = ky::Par<3>;
template <int AB> float run(int n, const uint8_t* seeds, int16_t* mat, hipEvent_t a, hipEvent_t b) {
  int jobs = n * 9;
  (void)hipEventRecord(a);
  ky::k90_sample_matrix<P, ky::AES_WG, AB><<<(jobs + ky::AES_WG - 1) / ky::AES_WG, ky::AES_WG>>>(jobs, seeds, 32, 1, mat);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  const int n = 524288;
  uint8_t* seeds; int16_t* mat;
  CK(hipMalloc(&seeds, n * 32)); CK(hipMalloc(&mat, (size_t)n * 9 * 512));
  { std::vector<uint32_t> h(n * 8); for (int i = 0; i < n * 8; i++) h[i] = i * 2654435761u; CK(hipMemcpy(seeds, h.data(), n * 32, hipMemcpyHostToDevice)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int r = 0; r < 4; r++) {
    float t0 = run<0>(n, seeds, mat, a, b), t1 = run<1>(n, seeds, mat, a, b), t2 = run<2>(n, seeds, mat, a, b), t3 = run<3>(n, seeds, mat, a, b), t4 = run<4>(n, seeds, mat, a, b);
    printf("round %d: full %.3f ms | no drain %.3f ms | keystream only %.3f ms | parse + drain only %.3f ms | parse only %.3f ms   (n=%d x 9 entries)\n", r, t0, t1, t2, t3, t4, n);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
