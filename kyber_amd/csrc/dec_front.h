// dec_front.h -- the front kernel of small decapsulations (batches up to FRONT_MAX_N instances, SHAKE sets).
//
// A decapsulation's first two stages do not depend on each other: the decryption arithmetic (k_dec_arith: ct, sk ->
// m') and the expansion of A^T from the rho stored in the secret key, which the re-encryption needs three launches
// later (ref/indcpa.c:318-328 and :270 as called from ref/kem.c:106-114).  On a chip that a small batch cannot fill
// they are ONE launch: the first workgroups decrypt, the rest sample, side by side (keccak_split.h, "front kernels",
// for the same idea in key generation and encapsulation).
//
// This is the ONE place where the poly-layout kernels (workgroup barriers for their LDS tables and transposes) and
// the sponge kernels (keccak.h's phase lock: raw s_barrier inside the permutation, sound only in workgroups that use
// no barrier for data) meet in a kernel.  They never meet in a WORKGROUP: the kernel is nothing but a branch on the
// workgroup index, taken the same way by every work-item of a workgroup, into one body or the other
// (tests/test_source_contracts.py checks exactly that shape).
#pragma once
#include "arith_kernels.h"
#include "kem_kernels.h"

namespace ky {

constexpr int DEC_FRONT_WG = 256;  // the poly layout's workgroup; the sampler body gets the same (four wavefronts, 31 KB of rows)

template <class P>
__global__ void __launch_bounds__(DEC_FRONT_WG) k_dec_front(int n, int nb_arith, const uint8_t* __restrict__ ct, const uint8_t* __restrict__ sk,
                                                            uint8_t* __restrict__ small, int16_t* __restrict__ mat,
                                                            uint32_t* __restrict__ tailq, uint32_t tail_cap) {
  if ((int)blockIdx.x < nb_arith) dec_arith_body<P>(blockIdx.x, n, ct, sk, P::SKB, small);
  else sample_matrix_body<P, DEC_FRONT_WG, 0>(blockIdx.x - nb_arith, n * P::NMAT, sk + 2 * P::PVB, P::SKB, 1, mat, tailq, tail_cap);
}

}  // namespace ky
