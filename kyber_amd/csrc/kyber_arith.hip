// kyber_arith.hip -- translation unit of the poly-layout (fp32 polynomial
// arithmetic) kernels and their launchers.  Built with -fno-slp-vectorize, see
// arith_kernels.h.
#include "arith_kernels.h"
#include "arith_launch.h"

namespace ky {

static inline unsigned nblocks(size_t items, unsigned per) { return (unsigned)((items + per - 1) / per); }

template <class P>
void launch_kp_arith(hipStream_t st, int n, const uint8_t* small, const int16_t* mat, const uint8_t* prf, uint8_t* pk, uint8_t* sk) {
  k_kp_arith<P><<<nblocks(n, G_INST), 256, 0, st>>>(n, small, mat, prf, pk, sk);
}
template <class P, bool CMP>
void launch_enc_arith(hipStream_t st, int n, const uint8_t* pk, size_t pk_stride, uint8_t* small, const int16_t* mat,
                      size_t mat_stride, const uint8_t* prf, uint8_t* ct) {
  k_enc_arith<P, CMP><<<nblocks(n, G_INST), 256, 0, st>>>(n, pk, pk_stride, small, mat, mat_stride, prf, ct);
}
template <class P>
void launch_dec_arith(hipStream_t st, int n, const uint8_t* ct, const uint8_t* sk, size_t sk_stride, uint8_t* small) {
  k_dec_arith<P><<<nblocks(n, G_INST), 256, 0, st>>>(n, ct, sk, sk_stride, small);
}

#define KY_INSTANTIATE(KK)                                                                                            \
  template void launch_kp_arith<Par<KK>>(hipStream_t, int, const uint8_t*, const int16_t*, const uint8_t*, uint8_t*, uint8_t*); \
  template void launch_enc_arith<Par<KK>, false>(hipStream_t, int, const uint8_t*, size_t, uint8_t*, const int16_t*, size_t,    \
                                                 const uint8_t*, uint8_t*);                                           \
  template void launch_enc_arith<Par<KK>, true>(hipStream_t, int, const uint8_t*, size_t, uint8_t*, const int16_t*, size_t,     \
                                                const uint8_t*, uint8_t*);                                            \
  template void launch_dec_arith<Par<KK>>(hipStream_t, int, const uint8_t*, const uint8_t*, size_t, uint8_t*);
KY_INSTANTIATE(2)
KY_INSTANTIATE(3)
KY_INSTANTIATE(4)

// grid-stride transform kernels: 16 workgroups per CU's worth of blocks, each looping over its share
static inline unsigned stride_grid(size_t npoly) {
  const unsigned full = nblocks(npoly, 8), cap = 256 * 16;
  return full < cap ? full : cap;
}
// U = 1: tools/microbench/ntt_ablate.hip measured U = 1, 2, 4 within 3 % of each other and of the
// bare load-convert-store loop (the kernel runs at its access pattern's copy rate); U = 1 keeps 8 waves per SIMD
void launch_poly_ntt(hipStream_t st, size_t npoly, int16_t* r) { k_poly_ntt<1><<<stride_grid(npoly), 256, 0, st>>>((long)npoly, r); }
void launch_poly_invntt(hipStream_t st, size_t npoly, int16_t* r) { k_poly_invntt<<<stride_grid(npoly), 256, 0, st>>>((long)npoly, r); }
void launch_poly_basemul(hipStream_t st, size_t npoly, int16_t* r, const int16_t* a, const int16_t* b) {
  k_poly_basemul<<<nblocks(npoly, 8), 256, 0, st>>>((long)npoly, r, a, b);
}
void launch_poly_mul_fused(hipStream_t st, size_t npoly, int16_t* r, const int16_t* a, const int16_t* b) {
  k_poly_mul_fused<<<stride_grid(npoly), 256, 0, st>>>((long)npoly, r, a, b);
}

}  // namespace ky
