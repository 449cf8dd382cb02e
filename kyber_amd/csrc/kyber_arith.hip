// kyber_arith.hip -- translation unit of the poly-layout (fp32 polynomial
// arithmetic) kernels and their launchers.  Built with -fno-slp-vectorize, see
// arith_kernels.h.
#include "arith_kernels.h"
#include "wpoly.h"
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

// grid-stride kernels of the int16 engine: 16 workgroups per CU's worth of blocks, each looping over its share
static inline unsigned stride_grid(size_t npoly) {
  const unsigned full = nblocks(npoly, 8), cap = 256 * 16;
  return full < cap ? full : cap;
}
// The forward transform and the canonical-output kernels run on the wide engine (wpoly.h): one
// launch-sized grid, a wavefront takes 4 U consecutive polynomials.  U and the register bound per
// kernel from tools/microbench/wpoly_rate.hip (profiles/r03_microbench_wpoly_rate.txt).
template <int OP, int U, int WAVES = 1>
static inline void launch_wide(hipStream_t st, size_t npoly, int16_t* r, const int16_t* a, const int16_t* b) {
  k_wpoly<OP, U, WAVES><<<nblocks(npoly, 16 * U), 256, 0, st>>>((long)npoly, r, a, b);
}
void launch_poly_ntt(hipStream_t st, size_t npoly, int16_t* r) { launch_wide<W_NTT, 2>(st, npoly, r, r, nullptr); }
void launch_poly_ntt_canon(hipStream_t st, size_t npoly, int16_t* r) { launch_wide<W_NTT_CANON, 2>(st, npoly, r, r, nullptr); }
void launch_poly_invntt_canon(hipStream_t st, size_t npoly, int16_t* r) { launch_wide<W_INVNTT_CANON, 2>(st, npoly, r, r, nullptr); }
void launch_poly_basemul_canon(hipStream_t st, size_t npoly, int16_t* r, const int16_t* a, const int16_t* b) {
  launch_wide<W_BASEMUL_CANON, 2>(st, npoly, r, a, b);
}
void launch_poly_mul_fused_canon(hipStream_t st, size_t npoly, int16_t* r, const int16_t* a, const int16_t* b) {
  launch_wide<W_MUL_CANON, 1, 5>(st, npoly, r, a, b);
}
void launch_poly_invntt(hipStream_t st, size_t npoly, int16_t* r) { k_poly_invntt<<<stride_grid(npoly), 256, 0, st>>>((long)npoly, r); }
void launch_poly_basemul(hipStream_t st, size_t npoly, int16_t* r, const int16_t* a, const int16_t* b) {
  k_poly_basemul<<<nblocks(npoly, 8), 256, 0, st>>>((long)npoly, r, a, b);
}
void launch_poly_mul_fused(hipStream_t st, size_t npoly, int16_t* r, const int16_t* a, const int16_t* b) {
  k_poly_mul_fused<<<stride_grid(npoly), 256, 0, st>>>((long)npoly, r, a, b);
}

// ---- single-function stages ----
bool launch_st_cbd(hipStream_t st, int eta, size_t npoly, int16_t* r, const uint8_t* buf) {
  if (eta == 2) k_st_cbd<2><<<nblocks(npoly, 8), 256, 0, st>>>((long)npoly, r, buf, 128);
  else if (eta == 3) k_st_cbd<3><<<nblocks(npoly, 8), 256, 0, st>>>((long)npoly, r, buf, 192);
  else return false;
  return true;
}
template <class P>
void launch_st_noise(hipStream_t st, size_t njobs, int nnonce, int n_eta1, int16_t* r, const uint8_t* prf) {
  k_st_noise<P><<<nblocks(njobs, 8), 256, 0, st>>>((long)njobs, nnonce, n_eta1, r, prf);
}
template void launch_st_noise<Par<2>>(hipStream_t, size_t, int, int, int16_t*, const uint8_t*);
template void launch_st_noise<Par<3>>(hipStream_t, size_t, int, int, int16_t*, const uint8_t*);
template void launch_st_noise<Par<4>>(hipStream_t, size_t, int, int, int16_t*, const uint8_t*);
void launch_st_tobytes(hipStream_t st, size_t npoly, uint8_t* out, const int16_t* a) {
  k_st_tobytes<<<nblocks(npoly, 8), 256, 0, st>>>((long)npoly, out, a);
}
void launch_st_frombytes(hipStream_t st, size_t npoly, int16_t* r, const uint8_t* in) {
  k_st_frombytes<<<nblocks(npoly, 8), 256, 0, st>>>((long)npoly, r, in);
}
bool launch_st_compress(hipStream_t st, int d, size_t npoly, uint8_t* out, const int16_t* a) {
  const unsigned nb = nblocks(npoly, 8);
  switch (d) {
    case 1: k_st_compress<1><<<nb, 256, 0, st>>>((long)npoly, out, a); return true;
    case 4: k_st_compress<4><<<nb, 256, 0, st>>>((long)npoly, out, a); return true;
    case 5: k_st_compress<5><<<nb, 256, 0, st>>>((long)npoly, out, a); return true;
    case 10: k_st_compress<10><<<nb, 256, 0, st>>>((long)npoly, out, a); return true;
    case 11: k_st_compress<11><<<nb, 256, 0, st>>>((long)npoly, out, a); return true;
  }
  return false;
}
bool launch_st_decompress(hipStream_t st, int d, size_t npoly, int16_t* r, const uint8_t* in) {
  const unsigned nb = nblocks(npoly, 8);
  switch (d) {
    case 1: k_st_decompress<1><<<nb, 256, 0, st>>>((long)npoly, r, in); return true;
    case 4: k_st_decompress<4><<<nb, 256, 0, st>>>((long)npoly, r, in); return true;
    case 5: k_st_decompress<5><<<nb, 256, 0, st>>>((long)npoly, r, in); return true;
    case 10: k_st_decompress<10><<<nb, 256, 0, st>>>((long)npoly, r, in); return true;
    case 11: k_st_decompress<11><<<nb, 256, 0, st>>>((long)npoly, r, in); return true;
  }
  return false;
}
bool launch_st_basemul_acc(hipStream_t st, int k, size_t nvec, int16_t* r, const int16_t* a, const int16_t* b) {
  const unsigned nb = nblocks(nvec, 8);
  switch (k) {
    case 2: k_st_basemul_acc<2><<<nb, 256, 0, st>>>((long)nvec, r, a, b); return true;
    case 3: k_st_basemul_acc<3><<<nb, 256, 0, st>>>((long)nvec, r, a, b); return true;
    case 4: k_st_basemul_acc<4><<<nb, 256, 0, st>>>((long)nvec, r, a, b); return true;
  }
  return false;
}

}  // namespace ky
