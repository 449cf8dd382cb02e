// matrix_ablate.hip -- where does k_sample_matrix spend its time?  Runs the
// product kernel (ABLATE=0) and ablations (1: no drain; 2: no parse, no drain;
// 3: no drain, staging stores at cursor-independent, bank-conflict-free addresses;
// 4: no drain, parse arithmetic without any staging store) interleaved in one
// process on the same seeds, at the product's batch (2^19 x 9 entries) and with its tail queue, so
// that entries still short after three blocks are parked as in the product.
#include "../../kyber_amd/csrc/kem_kernels.h"
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using P = ky::Par<3>;
uint32_t* g_tailq; uint32_t g_cap;
template <int AB> float run(int n, const uint8_t* seeds, int16_t* mat, hipEvent_t a, hipEvent_t b) {
  int jobs = n * 9;
  hipMemsetAsync(g_tailq, 0, 4);
  hipEventRecord(a);
  ky::k_sample_matrix<P, ky::SPONGE_WG, AB><<<(jobs + ky::SPONGE_WG - 1) / ky::SPONGE_WG, ky::SPONGE_WG>>>(jobs, seeds, 32, 1, mat, g_tailq, g_cap);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  const int n = 524288;
  g_cap = (uint32_t)ky::tailq_capacity((size_t)n * 9);
  CK(hipMalloc(&g_tailq, ky::tailq_bytes((size_t)n * 9)));
  uint8_t* seeds; int16_t* mat;
  CK(hipMalloc(&seeds, n * 32)); CK(hipMalloc(&mat, (size_t)n * 9 * 512));
  CK(hipMemset(seeds, 0x3c, n * 32));
  // distinct seeds: write index into first 4 bytes
  { std::vector<uint32_t> h(n * 8); for (int i = 0; i < n * 8; i++) h[i] = i * 2654435761u; CK(hipMemcpy(seeds, h.data(), n * 32, hipMemcpyHostToDevice)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int r = 0; r < 4; r++) {
    float t0 = run<0>(n, seeds, mat, a, b), t1 = run<1>(n, seeds, mat, a, b), t2 = run<2>(n, seeds, mat, a, b), t3 = run<3>(n, seeds, mat, a, b), t4 = run<4>(n, seeds, mat, a, b);
    printf("round %d: full %.3f ms | no drain %.3f ms | no drain, conflict-free staging stores %.3f ms | parse without stores %.3f ms | perm only %.3f ms   (n=%d x 9 entries)\n", r, t0, t1, t3, t4, t2, n);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
