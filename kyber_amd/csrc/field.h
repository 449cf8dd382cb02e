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

constexpr int Q = KY_Q;

// a * 2^-16 mod q, result in (-q, q); |a| < q * 2^15.      ref/reduce.c:16-23
__device__ __forceinline__ int mont_reduce(int a) {
  int m = (int)(short)__mul24(a, -3327);  // only the low 16 bits matter
  return (a - m * Q) >> 16;
}

// product of two int16-range values, Montgomery-reduced.    ref/ntt.c:68-70
__device__ __forceinline__ int fqmul(int a, int b) { return mont_reduce(__mul24(a, b)); }

// centred representative in [-(q-1)/2, (q-1)/2] of an int16-range value.  ref/reduce.c:35-42
__device__ __forceinline__ int barrett(int a) {
  int t = (20159 * a + (1 << 25)) >> 26;
  return a - t * Q;
}

// [0, q) representative of a value in (-q, q): what every serializer applies
// first (ref/poly.c:118-121, :28-30, :188-190; ref/polyvec.c:24-26).
__device__ __forceinline__ unsigned canon(int a) { return (unsigned)(a + ((a >> 31) & Q)); }

}  // namespace ky
