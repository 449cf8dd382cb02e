// arith_launch.h -- host-callable launchers of the poly-layout kernels.  The
// kernels live in kyber_arith.hip (own translation unit, own compiler flags);
// kyber_hip.hip sequences them through these functions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "params.h"

namespace ky {

// grids: one 256-thread workgroup per G_INST instances / per 8 polynomials
template <class P>
void launch_kp_arith(hipStream_t st, int n, const uint8_t* small, const int16_t* mat, const uint8_t* prf, uint8_t* pk, uint8_t* sk);
template <class P, bool CMP>
void launch_enc_arith(hipStream_t st, int n, const uint8_t* pk, size_t pk_stride, uint8_t* small, const int16_t* mat,
                      size_t mat_stride, const uint8_t* prf, uint8_t* ct);
template <class P>
void launch_dec_arith(hipStream_t st, int n, const uint8_t* ct, const uint8_t* sk, size_t sk_stride, uint8_t* small);

void launch_poly_ntt(hipStream_t st, size_t npoly, int16_t* r);
void launch_poly_invntt(hipStream_t st, size_t npoly, int16_t* r);
void launch_poly_basemul(hipStream_t st, size_t npoly, int16_t* r, const int16_t* a, const int16_t* b);
void launch_poly_mul_fused(hipStream_t st, size_t npoly, int16_t* r, const int16_t* a, const int16_t* b);
// canonical-output kernels of the wide engine (wpoly.h)
void launch_poly_ntt_canon(hipStream_t st, size_t npoly, int16_t* r);
void launch_poly_invntt_canon(hipStream_t st, size_t npoly, int16_t* r);
void launch_poly_basemul_canon(hipStream_t st, size_t npoly, int16_t* r, const int16_t* a, const int16_t* b);
void launch_poly_mul_fused_canon(hipStream_t st, size_t npoly, int16_t* r, const int16_t* a, const int16_t* b);

// single-function stages (tests/test_gpu_stages.py); eta / d / k outside the supported sets return false
bool launch_st_cbd(hipStream_t st, int eta, size_t npoly, int16_t* r, const uint8_t* buf);
template <class P>
void launch_st_noise(hipStream_t st, size_t njobs, int nnonce, int n_eta1, int16_t* r, const uint8_t* prf);
void launch_st_tobytes(hipStream_t st, size_t npoly, uint8_t* out, const int16_t* a);
void launch_st_frombytes(hipStream_t st, size_t npoly, int16_t* r, const uint8_t* in);
bool launch_st_compress(hipStream_t st, int d, size_t npoly, uint8_t* out, const int16_t* a);
bool launch_st_decompress(hipStream_t st, int d, size_t npoly, int16_t* r, const uint8_t* in);
bool launch_st_basemul_acc(hipStream_t st, int k, size_t nvec, int16_t* r, const int16_t* a, const int16_t* b);

}  // namespace ky
