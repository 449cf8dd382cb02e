// field.h -- arithmetic in Z_3329 on 32-bit VALU lanes.
//
// Coefficients travel as int16 in memory/LDS and as sign-extended int32 in
// registers; every function returns exactly the int16 value the reference
// would hold at the same point, so intermediate stages compare bit for bit.
//
// Replaces: ref/reduce.c:16-23 montgomery_reduce, :35-42 barrett_reduce,
// ref/ntt.c:68-70 fqmul, ref/ntt.c:39-56 zetas.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "consts_gen.h"

namespace ky {

__constant__ const int16_t ZETAS[128] = KY_ZETAS_INIT;
constexpr int16_t ZETAS_HOST[128] = KY_ZETAS_INIT;  // for compile-time twiddle pairs

constexpr int Q = KY_Q;

// 16-bit multiplier forms the compiler does not select on its own.  They read
// only the low 16 bits of their 16-bit operands (as signed for the mad), so
// int16-range values need no sign extension or masking around them:
//   v_mad_i32_i16  d = (int16)a * (int16)b + c      half rate
//   v_mul_lo_u16   d = low16(a * b), upper half 0    full rate
__device__ __forceinline__ int mad_i16(int a, int b, int c) {
  int d;
  asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ int mul_i16(int a, int b) {
  int d;
  asm("v_mad_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ int mul_lo16(int a, int b) {
  int d;
  asm("v_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// low16(a.lo16 * w.hi16)
__device__ __forceinline__ int mul_lo16_w1(int a, int w) {
  int d;
  asm("v_mul_lo_u16_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(d) : "v"(a), "v"(w));
  return d;
}

// a * 2^-16 mod q, result in (-q, q); |a| < q * 2^15.      ref/reduce.c:16-23
// t = int16(a * q^-1); r = (a - t q) >> 16, six issue slots on gfx950.
__device__ __forceinline__ int mont_reduce(int a) {
  const int m = mul_lo16(a, KY_QINV);
  return mad_i16(m, -Q, a) >> 16;
}

// product of two int16-range values, Montgomery-reduced.    ref/ntt.c:68-70
__device__ __forceinline__ int fqmul(int a, int b) { return mont_reduce(mul_i16(a, b)); }

// The same with a precomputed twiddle pair w = zeta | (zeta * q^-1 mod 2^16) << 16:
// low16(b * zeta * q^-1) = low16(b * w.hi), so the quotient does not wait for the product.
__device__ __forceinline__ int fqmul_tw(int b, int w) {
  const int p = mul_i16(w, b);
  const int m = mul_lo16_w1(b, w);
  return mad_i16(m, -Q, p) >> 16;
}
__host__ __device__ constexpr int tw_pack(int zeta) {
  return (int)(((unsigned)zeta & 0xFFFFu) | ((unsigned)(zeta * KY_QINV) & 0xFFFFu) << 16);
}

// centred representative in [-(q-1)/2, (q-1)/2] of an int16-range value.  ref/reduce.c:35-42
__device__ __forceinline__ int barrett(int a) {
  int t = (20159 * a + (1 << 25)) >> 26;
  return a - t * Q;
}

// [0, q) representative of a value in (-q, q): what every serializer applies
// first (ref/poly.c:118-121, :28-30, :188-190; ref/polyvec.c:24-26).
__device__ __forceinline__ unsigned canon(int a) { return (unsigned)(a + ((a >> 31) & Q)); }

}  // namespace ky
