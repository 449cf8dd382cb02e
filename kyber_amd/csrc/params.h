// params.h -- compile-time parameter sets and workspace constants shared by the
// sponge-layout kernels (kem_kernels.h) and the poly-layout kernels (arith_kernels.h).
#pragma once

namespace ky {

template <int K_>
struct Par {
  static constexpr int K = K_;
  static constexpr int ETA1 = (K_ == 2) ? 3 : 2;          // ref/params.h:42-54
  static constexpr int DU = (K_ == 4) ? 11 : 10;
  static constexpr int DV = (K_ == 4) ? 5 : 4;
  static constexpr int PVB = 384 * K_;                     // polyvec bytes
  static constexpr int PKB = PVB + 32;                     // ref/params.h:63
  static constexpr int SKB = 2 * PVB + 96;                 // ref/params.h:65
  static constexpr int CTB = 32 * (DU * K_ + DV);          // ref/params.h:66
  static constexpr int UB = 32 * DU;                       // one compressed u poly
  static constexpr int NMAT = K_ * K_;
  static constexpr int PRF_STRIDE = 64 * ETA1;             // bytes per PRF job slot (>= 128)
};

// workspace "small" record, 128 bytes per instance
//   enc/dec: [0,32) m   [32,64) Kbar   [64,96) r (coins)   [96,100) fail flag
//   keypair: [0,32) rho [32,64) sigma
constexpr int SMALL = 128;
constexpr int G_INST = 8;  // instances per poly-layout workgroup

}  // namespace ky
