// aes_rate.hip -- microbenchmark: AES-256-CTR blocks per second out of the LDS tables of aes.h, one
// cipher context per work-item, as k90_sample_matrix / k90_prf run it.  Prices the two resources
// of a table round separately: the LDS pipe (16 ds_read_b32 per round and wavefront) and the VALU
// (addresses, one rotate and two 3-input xors per column).
//   MODE 0  aes256ctr_block as the product calls it, one block after the other
//   MODE 1, 4, 5  two, three, four blocks of one stream advanced round by round (look-ups of one beside the logic of the others)
//   MODE 2  the look-ups alone: 16 per round, addresses one v_bitop3 each from the previous round's sum
//   MODE 3  the VALU work alone: the round's instructions with the look-ups replaced by moves
#include "../../kyber_amd/csrc/aes.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using namespace ky;

template <int WG, int MODE>
__global__ void __launch_bounds__(WG) k_aes(const uint32_t* __restrict__ keys, uint32_t* __restrict__ out, int nblk, unsigned long long* clk) {
  KY_AES_TABLE(tab);
  aes_table_init(tab);
  const AesLds L = aes_lds(tab);
  const long i = (long)blockIdx.x * WG + threadIdx.x;
  u64p key[4];
  uint32_t rk[60];
#pragma unroll
  for (int w = 0; w < 4; w++) key[w] = u64p{keys[i * 8 + 2 * w], keys[i * 8 + 2 * w + 1]};
  aes256_expand(L, key, rk);
  const AesCtr cx = aes256ctr_init(L, rk, (uint32_t)i & 0xFFFFu);
  uint32_t acc[4] = {0, 0, 0, 0};
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  if constexpr (MODE == 0) {
#pragma unroll 1
    for (int b = 0; b < nblk; b++) {
      uint32_t o[4];
      aes256ctr_block(L, rk, cx, (uint32_t)b, o);
#pragma unroll
      for (int w = 0; w < 4; w++) acc[w] ^= o[w];
    }
  } else if constexpr (MODE == 1) {
#pragma unroll 1
    for (int b = 0; b < nblk; b += 2) {
      uint32_t o[2][4];
      aes256ctr_blocks<2>(L, rk, cx, (uint32_t)b, o);
#pragma unroll
      for (int w = 0; w < 4; w++) acc[w] = bitop3<0x96>(acc[w], o[0][w], o[1][w]);
    }
  } else if constexpr (MODE == 4) {
#pragma unroll 1
    for (int b = 0; b < nblk - 2; b += 3) {   // 30 blocks of the 32, as run() counts them
      uint32_t o[3][4];
      aes256ctr_blocks<3>(L, rk, cx, (uint32_t)b, o);
#pragma unroll
      for (int w = 0; w < 4; w++) acc[w] ^= bitop3<0x96>(o[2][w], o[0][w], o[1][w]);
    }
  } else if constexpr (MODE == 5) {
#pragma unroll 1
    for (int b = 0; b < nblk; b += 4) {
      uint32_t o[4][4];
      aes256ctr_blocks<4>(L, rk, cx, (uint32_t)b, o);
#pragma unroll
      for (int w = 0; w < 4; w++) acc[w] ^= bitop3<0x96>(o[2][w], o[0][w], o[1][w]) ^ o[3][w];
    }
  } else if constexpr (MODE == 2) {
    uint32_t s0 = rk[8], s1 = rk[9], s2 = rk[10], s3 = rk[11];
#pragma unroll 1
    for (int b = 0; b < nblk; b++) {
#pragma unroll
      for (int r = 1; r < 14; r++) {
        uint32_t t[4];
        const uint32_t q[4] = {s0, s1, s2, s3};
#pragma unroll
        for (int c = 0; c < 4; c++) {
          // sixteen different masks, so that no two look-ups of a round share an address
          const uint32_t x0 = L.at(bitop3<0xEA>(q[c], 0xFF00u - ((4 * c) << 8), L.off[0])), x1 = L.at(bitop3<0xEA>(q[(c + 1) & 3], 0xFF00u - ((4 * c + 1) << 8), L.off[1]));
          const uint32_t x2 = L.at(bitop3<0xEA>(q[(c + 2) & 3], 0xFF00u - ((4 * c + 2) << 8), L.off[0])), x3 = L.at(bitop3<0xEA>(q[(c + 3) & 3], 0xFF00u - ((4 * c + 3) << 8), L.off[1]));
          t[c] = bitop3<0x96>(x0, x1, x2) ^ x3 ^ rk[4 * r + c];
        }
        s0 = t[1]; s1 = t[2]; s2 = t[3]; s3 = t[0];
      }
      acc[0] ^= s0; acc[1] ^= s1; acc[2] ^= s2; acc[3] ^= s3;
    }
  } else {
    uint32_t s0 = rk[8], s1 = rk[9], s2 = rk[10], s3 = rk[11];
#pragma unroll 1
    for (int b = 0; b < nblk; b++) {
#pragma unroll
      for (int r = 1; r < 14; r++) {
        uint32_t t[4];
        const uint32_t q[4] = {s0, s1, s2, s3};
#pragma unroll
        for (int c = 0; c < 4; c++) {
          uint32_t a0 = __builtin_amdgcn_perm(q[c], L.off[0], 0x03020400u), a1 = bitop3<0xEA>(q[(c + 1) & 3], 0xFF00u, L.off[0]);
          uint32_t a2 = __builtin_amdgcn_perm(q[(c + 2) & 3], L.off[1], 0x03020600u), a3 = __builtin_amdgcn_perm(q[(c + 3) & 3], L.off[1], 0x03020700u);
          asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));   // where the look-ups would be
          t[c] = bitop3<0x96>(a0, a2, rotl32(bitop3<0x96>(a1, a3, rk[4 * r + c]), 8));
        }
        s0 = t[0]; s1 = t[1]; s2 = t[2]; s3 = t[3];
      }
      acc[0] ^= s0; acc[1] ^= s1; acc[2] ^= s2; acc[3] ^= s3;
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  out[i] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int WG, int MODE>
int run(const char* name, const uint32_t* keys, uint32_t* out, unsigned long long* clk, int n, int nblk, size_t pad) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < 4; r++) {
    CK(hipEventRecord(a));
    k_aes<WG, MODE><<<n / WG, WG, pad>>>(keys, out, nblk, clk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  std::vector<unsigned long long> h(2 * (n / WG));
  CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
  double cs = 0, ws = 0;
  for (size_t i = 0; i < h.size(); i += 2) { cs += h[i]; ws += h[i + 1]; }
  int wall_khz = 0; CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
  const double mhz = cs / ws * wall_khz / 1e3;
  const double blocks = (double)n * (MODE == 4 ? nblk / 3 * 3 : nblk);
  // look-ups per block: 197 in counter mode with the first two rounds folded (aes.h), 13 x 16 in MODE 2, none in MODE 3
  const int per_block = MODE == 2 ? 208 : MODE == 3 ? 0 : 197;
  printf("%-44s %7.3f ms  %7.2f G blocks/s  %6.1f G look-ups/s per CU (peak 32/clk = %5.1f)  %4.0f MHz\n", name, best, blocks / best / 1e6,
         blocks * per_block / best / 1e6 / 256, 32 * mhz / 1e3, mhz);
  return 0;
}

int main() {
  const int n = 1 << 22, nblk = 32;   // 16 workgroups of 1024 per CU
  uint32_t *keys, *out; unsigned long long* clk;
  CK(hipMalloc(&keys, (size_t)n * 32)); CK(hipMalloc(&out, (size_t)n * 4)); CK(hipMalloc(&clk, (size_t)(n / 64) * 16));
  std::vector<uint32_t> h((size_t)n * 8);
  uint32_t x = 0x9E3779B9u;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = x; }
  CK(hipMemcpy(keys, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  if (run<1024, 0>("(warm-up)", keys, out, clk, n, nblk, 0)) return 1;
  if (run<1024, 0>("product block, 4 waves/SIMD (WG 1024)", keys, out, clk, n, nblk, 0)) return 1;
  if (run<512, 0>("product block, 2+2 waves/SIMD (2 x WG 512)", keys, out, clk, n, nblk, 0)) return 1;
  if (run<256, 0>("product block, 1+1 waves/SIMD (2 x WG 256)", keys, out, clk, n, nblk, 0)) return 1;
  if (run<1024, 1>("two blocks round by round, 4 waves/SIMD", keys, out, clk, n, nblk, 0)) return 1;
  if (run<512, 1>("two blocks round by round, 2+2 waves/SIMD", keys, out, clk, n, nblk, 0)) return 1;
  if (run<1024, 4>("three blocks round by round, 4 waves/SIMD", keys, out, clk, n, nblk, 0)) return 1;
  if (run<1024, 5>("four blocks round by round, 4 waves/SIMD", keys, out, clk, n, nblk, 0)) return 1;
  if (run<512, 5>("four blocks round by round, 2+2 waves/SIMD", keys, out, clk, n, nblk, 0)) return 1;
  if (run<1024, 2>("look-ups alone, 4 waves/SIMD", keys, out, clk, n, nblk, 0)) return 1;
  if (run<512, 2>("look-ups alone, 2+2 waves/SIMD", keys, out, clk, n, nblk, 0)) return 1;
  if (run<1024, 3>("VALU work alone, 4 waves/SIMD", keys, out, clk, n, nblk, 0)) return 1;
  // last: raising the limit also changes how the unpadded launches of that kernel run
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_aes<512, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
  if (run<512, 0>("product block, 2 waves/SIMD (1 x WG 512)", keys, out, clk, n, nblk, 48 * 1024)) return 1;
  return 0;
}
