// aes_bitslice.hip -- the question VERDICT r02 put to the 90s sampler: LDS tables (aes.h, what the
// product runs: 197 ds_read_b32 per block, bound by the LDS pipe) or a bitsliced AES on the VALU?
//
// Bitsliced form measured here: one work-item encrypts the 32 counter blocks of one stream at once --
// exactly a matrix entry's first squeeze (ref/indcpa.c:160-162: 8 x 64 bytes).  The state is 128 bit
// planes of 32 bits (plane 32c + 8r + b = bit b of the byte in row r, column c, one bit per block), so
// ShiftRows is renaming, MixColumns is xors, SubBytes is 16 S-box circuits on 8 planes each
// (aes_bs_sbox.h, generated and verified by tools/gen_aes_bitslice.py: 171 gates, which the compiler
// folds into v_bitop3_b32 where three inputs meet) and AddRoundKey spreads each key bit over a plane
// (v_bfe_i32).  The counter's five low bits are the constant planes 0xAAAAAAAA ... 0xFFFF0000, every
// other input plane is all-zeros or all-ones.  At the end four 32 x 32 bit transposes turn the planes
// back into the 32 keystream blocks the rejection sampler parses.
//
// Keys: 16 different expanded keys in global memory, work-item i uses key i % 16 and nonce i & 0xFFFF
// (in the product every work-item has its own key; 240 bytes of round keys per work-item cannot live in
// registers next to 128 planes, so they are fetched per round here too: four dwords per lane and round).
// Checked against a plain byte-oriented AES-256 (FIPS 197) on the host for the first work-items.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../kyber_amd/csrc/consts_gen.h"
#include "aes_bs_sbox.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int NKEY = 16;

__device__ __forceinline__ uint32_t spread(uint32_t w, int k) { return (uint32_t)((int32_t)(w << (31 - k)) >> 31); }

__device__ __forceinline__ void add_round_key(uint32_t (&st)[128], uint4 k) {
  const uint32_t w[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int b = 0; b < 32; b++) st[32 * c + b] ^= spread(w[c], b);
}
__device__ __forceinline__ void sub_bytes(uint32_t (&st)[128]) {
#pragma unroll
  for (int i = 0; i < 16; i++) {
    uint32_t x[8];
#pragma unroll
    for (int b = 0; b < 8; b++) x[b] = st[8 * i + b];
    bs_sbox(x);
#pragma unroll
    for (int b = 0; b < 8; b++) st[8 * i + b] = x[b];
  }
}
__device__ __forceinline__ void shift_rows(uint32_t (&st)[128]) {  // row r moves left by r columns: renaming
  uint32_t o[128];
#pragma unroll
  for (int i = 0; i < 128; i++) o[i] = st[i];
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int b = 0; b < 8; b++) st[32 * c + 8 * r + b] = o[32 * ((c + r) & 3) + 8 * r + b];
}
__device__ __forceinline__ void mix_columns(uint32_t (&st)[128]) {  // out_r = a_r ^ (a0^a1^a2^a3) ^ xtime(a_r ^ a_{r+1})
#pragma unroll
  for (int c = 0; c < 4; c++) {
    uint32_t a[4][8], all[8];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int b = 0; b < 8; b++) a[r][b] = st[32 * c + 8 * r + b];
#pragma unroll
    for (int b = 0; b < 8; b++) all[b] = a[0][b] ^ a[1][b] ^ a[2][b] ^ a[3][b];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      uint32_t t[8];
#pragma unroll
      for (int b = 0; b < 8; b++) t[b] = a[r][b] ^ a[(r + 1) & 3][b];
#pragma unroll
      for (int b = 0; b < 8; b++) {
        uint32_t x = b ? t[b - 1] : t[7];
        if (b == 1 || b == 3 || b == 4) x ^= t[7];  // reduction by x^8 + x^4 + x^3 + x + 1
        st[32 * c + 8 * r + b] = a[r][b] ^ all[b] ^ x;
      }
    }
  }
}
// 32 x 32 bit transpose, bit i of a[k] <-> bit k of a[i]
__device__ __forceinline__ void transpose32(uint32_t* a) {
  constexpr uint32_t M[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
#pragma unroll
  for (int s = 0; s < 5; s++) {
    const int j = 16 >> s;
#pragma unroll
    for (int k = 0; k < 32; k++) {
      if (k & j) continue;
      const uint32_t t = ((a[k] >> j) ^ a[k | j]) & M[s];
      a[k | j] ^= t;
      a[k] ^= t << j;
    }
  }
}

// CHECK: write the 32 x 16 keystream bytes of every work-item; otherwise fold them into one word
template <bool CHECK, int WAVES = 1>
__global__ void __launch_bounds__(256, WAVES) k_aes_bs(const uint4* __restrict__ rkeys, uint32_t* __restrict__ out, uint32_t ctr0) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const uint4* rk = rkeys + (i % NKEY) * 15;
  const uint32_t nonce = (uint32_t)i & 0xFFFFu;
  uint32_t st[128];
#pragma unroll
  for (int p = 0; p < 128; p++) st[p] = 0;
#pragma unroll
  for (int b = 0; b < 16; b++) st[b] = spread(nonce, b);  // bytes 0 and 1: column 0, rows 0 and 1
  // the counter is big-endian in bytes 12..15 = column 3, rows 0..3: bit k of ctr0 + j at plane 96 + 8 (3 - k/8) + k%8
  constexpr uint32_t LOW[5] = {0xAAAAAAAAu, 0xCCCCCCCCu, 0xF0F0F0F0u, 0xFF00FF00u, 0xFFFF0000u};  // ctr0 is a multiple of 32
#pragma unroll
  for (int k = 0; k < 32; k++) st[96 + 8 * (3 - k / 8) + k % 8] = k < 5 ? LOW[k] : spread(ctr0, k);
  add_round_key(st, rk[0]);
#pragma unroll 1
  for (int r = 1; r < 14; r++) {
    sub_bytes(st);
    shift_rows(st);
    mix_columns(st);
    add_round_key(st, rk[r]);
  }
  sub_bytes(st);
  shift_rows(st);
  add_round_key(st, rk[14]);
#pragma unroll
  for (int c = 0; c < 4; c++) transpose32(st + 32 * c);  // st[32c + j] = word c of block j
  if constexpr (CHECK) {
#pragma unroll
    for (int j = 0; j < 32; j++)
      *reinterpret_cast<uint4*>(out + (i * 32 + j) * 4) = make_uint4(st[j], st[32 + j], st[64 + j], st[96 + j]);
  } else {
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < 128; p++) acc ^= st[p];
    out[i] = acc;
  }
}

// ---------------------------------------------------------------- host model: FIPS 197, byte by byte
static const uint8_t SBOX[256] = KY_AES_SBOX_INIT;
static uint8_t xt(uint8_t a) { return (uint8_t)((a << 1) ^ ((a >> 7) * 0x1B)); }
static void expand256(const uint8_t key[32], uint8_t rk[15][16]) {
  uint8_t w[60][4];
  memcpy(w, key, 32);
  uint8_t rcon = 1;
  for (int i = 8; i < 60; i++) {
    uint8_t t[4] = {w[i - 1][0], w[i - 1][1], w[i - 1][2], w[i - 1][3]};
    if (i % 8 == 0) {
      const uint8_t u[4] = {(uint8_t)(SBOX[t[1]] ^ rcon), SBOX[t[2]], SBOX[t[3]], SBOX[t[0]]};
      memcpy(t, u, 4);
      rcon = xt(rcon);
    } else if (i % 8 == 4) {
      for (int k = 0; k < 4; k++) t[k] = SBOX[t[k]];
    }
    for (int k = 0; k < 4; k++) w[i][k] = w[i - 8][k] ^ t[k];
  }
  memcpy(rk, w, 240);
}
static void encrypt(const uint8_t rk[15][16], uint8_t s[16]) {  // s[4c + r]
  for (int k = 0; k < 16; k++) s[k] ^= rk[0][k];
  for (int r = 1; r <= 14; r++) {
    uint8_t t[16];
    for (int c = 0; c < 4; c++)
      for (int q = 0; q < 4; q++) t[4 * c + q] = SBOX[s[4 * ((c + q) & 3) + q]];
    if (r < 14)
      for (int c = 0; c < 4; c++) {
        const uint8_t* a = t + 4 * c;
        const uint8_t all = a[0] ^ a[1] ^ a[2] ^ a[3];
        for (int q = 0; q < 4; q++) s[4 * c + q] = a[q] ^ all ^ xt(a[q] ^ a[(q + 1) & 3]);
      }
    else memcpy(s, t, 16);
    for (int k = 0; k < 16; k++) s[k] ^= rk[r][k];
  }
}

int main() {
  const int n = 1 << 22;  // 2^22 streams of 32 blocks, as aes_rate
  std::vector<uint8_t> keys(NKEY * 32);
  uint32_t x = 0x9E3779B9u;
  for (auto& v : keys) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = (uint8_t)(x >> 11); }
  static uint8_t rk[NKEY][15][16];
  for (int k = 0; k < NKEY; k++) expand256(&keys[32 * k], rk[k]);
  {  // FIPS 197 appendix C.3 as a self-check of the host model
    uint8_t key[32], pt[16], sched[15][16];
    for (int k = 0; k < 32; k++) key[k] = (uint8_t)k;
    for (int k = 0; k < 16; k++) pt[k] = (uint8_t)(0x11 * k);
    expand256(key, sched);
    encrypt(sched, pt);
    if (pt[0] != 0x8e || pt[1] != 0xa2 || pt[15] != 0x89) { printf("host AES model wrong\n"); return 2; }
  }
  uint4* drk; uint32_t* dout;
  CK(hipMalloc(&drk, sizeof rk)); CK(hipMemcpy(drk, rk, sizeof rk, hipMemcpyHostToDevice));
  CK(hipMalloc(&dout, (size_t)n * 4 > (size_t)4096 * 512 ? (size_t)n * 4 : (size_t)4096 * 512));
  const uint32_t ctr0 = 0;
  // ---- correctness: 4096 streams, blocks 0..31 and a later window (counter 0x01020300: every counter byte non-zero)
  for (uint32_t c0 : {0u, 0x01020300u}) {
    const int nc = 4096;
    k_aes_bs<true><<<nc / 256, 256>>>(drk, dout, c0);
    std::vector<uint8_t> got((size_t)nc * 512);
    CK(hipMemcpy(got.data(), dout, got.size(), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i : {0, 1, 2, 15, 16, 17, 255, 256, 1000, 4095})
      for (int j = 0; j < 32; j++) {
        uint8_t s[16] = {0};
        s[0] = (uint8_t)i; s[1] = (uint8_t)(i >> 8);
        const uint32_t c = c0 + (uint32_t)j;
        s[12] = (uint8_t)(c >> 24); s[13] = (uint8_t)(c >> 16); s[14] = (uint8_t)(c >> 8); s[15] = (uint8_t)c;
        encrypt(rk[i % NKEY], s);
        if (memcmp(s, &got[((size_t)i * 32 + j) * 16], 16)) bad++;
      }
    printf("bitsliced AES-256-CTR, counters 0x%08x..+31: %s\n", c0, bad ? "MISMATCH" : "ok (equal to the FIPS 197 model)");
    if (bad) return 2;
  }
  // ---- rate
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto rate = [&](const char* name, auto launch) {
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
      (void)hipEventRecord(a);
      launch();
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    printf("%-66s %7.3f ms  %7.2f G blocks/s\n", name, best, (double)n * 32 / best / 1e6);
  };
  printf("S-box circuit: %d gates (tower field, generated); the published minimum is 113\n", KY_BS_SBOX_GATES);
  rate("bitsliced, 32 blocks per work-item, 1 wavefront per SIMD (276 VGPR)", [&] { k_aes_bs<false, 1><<<n / 256, 256>>>(drk, dout, ctr0); });
  rate("bitsliced, 32 blocks per work-item, 2 wavefronts per SIMD (256 VGPR, 13 spilled)", [&] { k_aes_bs<false, 2><<<n / 256, 256>>>(drk, dout, ctr0); });
  // (a bound of 3 wavefronts per SIMD = 168 registers spills 335 of them)
  return 0;
}
