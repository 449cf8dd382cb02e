// aes.h -- AES-256 in counter mode for the -90s parameter sets, one cipher
// context per work-item.
//
// The reference runs a bitsliced constant-time AES four blocks at a time
// (ref/aes256ctr.c:339-520).  On gfx950 the natural form is the classic
// 32-bit "T-table" round out of LDS.  Two tables are kept: T0[x] = (2*S[x], S[x],
// S[x], 3*S[x]) and T2[x] = T0[x] rotated by 16 bits, so that a column
//     T0[a.b0] ^ rotl8 T0[b.b1] ^ rotl16 T0[c.b2] ^ rotl24 T0[d.b3] ^ k
//   = T0[a.b0] ^ T2[c.b2] ^ rotl8(T0[b.b1] ^ T2[d.b3] ^ rotr8 k)
// costs one rotate (v_alignbit_b32, a half-rate instruction) instead of three; the
// middle round keys are stored rotated.  Each table is stored 32 times, interleaved
// (entry x of table t, copy c at dword 64 x + 32 t + c), and lane l reads copy
// l % 32.  ds_read_b32 serves a wavefront as two groups of 32 lanes with bank =
// dword address mod 32 (MI355X_MICROARCH.md, LDS table), so every lane of a group
// owns one bank whatever its (key- and data-dependent) index is: the look-ups are
// free of bank conflicts, which also makes their timing independent of the secret
// PRF keys.  With 256 bytes per entry the byte address of a look-up is
//     x << 8 | t << 7 | c << 2   (+ the table's base, a multiple of 64 KiB),
// i.e. byte 1 = the state byte and bytes 0, 2, 3 = a per-lane constant: one
// v_perm_b32 builds it from the state word (one v_bitop3_b32 when the state byte
// already is byte 1), where extract + scale + add took two to three instructions.
// The look-ups of a block are 16 per round on the LDS pipe (two cycles each per
// wavefront, four SIMDs sharing the pipe), which is what bounds the 90s sampler.
//
// Counter layout restated from ref/aes256ctr.c:530-543,565-577: block input =
// 12-byte nonce || 32-bit big-endian block counter.
#pragma once
#include "keccak.h"
#include "sha2.h"

namespace ky {

__constant__ const uint8_t AES_SBOX[256] = KY_AES_SBOX_INIT;

constexpr int AES_TCOPIES = 32;                  // ds_read_b32 banks are (a/4) mod 32 per 32-lane group: one private bank per lane
constexpr int AES_TABLE_DW = 256 * 2 * AES_TCOPIES;  // two tables: 64 KiB of LDS
// Lanes per workgroup of the AES kernels.  One table fits a CU, so a CU runs one workgroup: 1024 lanes
// (four wavefronts per SIMD) when there is work for every CU, smaller workgroups on more CUs when not.
constexpr int AES_WG = 1024, AES_WG_MID = 512, AES_WG_SMALL = 256;
inline int aes_wg(long jobs, int cus) { return jobs <= (long)cus * AES_WG_SMALL ? AES_WG_SMALL : jobs <= (long)cus * AES_WG_MID ? AES_WG_MID : AES_WG; }

// Keystream blocks a work-item advances together (aes256ctr_blocks).  Pairs are 2 % faster where the AES
// rounds run alone (tools/microbench/aes_rate; 11 % before aes_round issued a round's look-ups back to
// back), but inside the kernels of the KEM one block after the other is as fast or faster
// (tools/lib_ab.sh, -DKY_AES_NB=2: k90_sample_matrix 3.03 against 2.99 ms).
#ifndef KY_AES_NB
#define KY_AES_NB 1
#endif
constexpr int AES_NB = KY_AES_NB;
static_assert(AES_NB == 1 || AES_NB == 2, "the kernels produce keystream singly or in pairs");

__device__ __forceinline__ uint32_t aes_xtime(uint32_t s) { return ((s << 1) ^ ((s & 0x80u) ? 0x1Bu : 0u)) & 0xFFu; }

// every work-item of the workgroup takes part; ends with a barrier
__device__ __forceinline__ void aes_table_init(uint32_t* tab) {
  for (int x = threadIdx.x; x < 256; x += blockDim.x) {
    uint32_t s = AES_SBOX[x], s2 = aes_xtime(s), s3 = s2 ^ s;
    uint32_t t = s2 | s << 8 | s << 16 | s3 << 24;
    uint32_t u = t << 16 | t >> 16;
#pragma unroll
    for (int c = 0; c < AES_TCOPIES; c++) {
      tab[x * 2 * AES_TCOPIES + c] = t;
      tab[x * 2 * AES_TCOPIES + AES_TCOPIES + c] = u;
    }
  }
  __syncthreads();
}

// The table is aligned to its own size (64 KiB) in LDS: off[t] holds every address bit
// except those of the entry number (base | t << 7 | 4 * (l % 32)), with byte 1 zero.
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
struct AesLds {
  uint32_t off[2];
  __device__ __forceinline__ uint32_t at(uint32_t a) const { return *(lds_cu32*)(uintptr_t)a; }
  // address of table T's entry (byte K of s)
  template <int K, int T>
  __device__ __forceinline__ uint32_t addr(uint32_t s) const {
    if constexpr (K == 1) return bitop3<0xEA>(s, 0xFF00u, off[T]);                  // (s & 0xFF00) | off
    else return __builtin_amdgcn_perm(s, off[T], 0x03020000u | (uint32_t)(4 + K) << 8);  // [off.b0, s.bK, off.b2, off.b3]
  }
  template <int K, int T>
  __device__ __forceinline__ uint32_t look(uint32_t s) const { return at(addr<K, T>(s)); }
  __device__ __forceinline__ uint32_t sbox(uint32_t x) const { return (at(x << 8 | off[0]) >> 8) & 0xFFu; }
};
#define KY_AES_TABLE(name) __shared__ __attribute__((aligned(4 * AES_TABLE_DW))) uint32_t name[AES_TABLE_DW]
__device__ __forceinline__ AesLds aes_lds(const uint32_t* tab) {
  const uint32_t b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)tab + 4u * (threadIdx.x & (AES_TCOPIES - 1));
  return AesLds{{b, b + 4u * AES_TCOPIES}};
}

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }

__device__ __forceinline__ uint32_t aes_subword(const AesLds& L, uint32_t w) {
  return L.sbox(w & 0xFF) | L.sbox((w >> 8) & 0xFF) << 8 | L.sbox((w >> 16) & 0xFF) << 16 | L.sbox(w >> 24) << 24;
}

// one output column of a middle round; kr = the round key word rotated right by 8
__device__ __forceinline__ uint32_t aes_col(const AesLds& L, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t kr) {
  const uint32_t x0 = L.look<0, 0>(a), x1 = L.look<1, 0>(b), y2 = L.look<2, 1>(c), y3 = L.look<3, 1>(d);
  return bitop3<0x96>(x0, y2, rotl32(bitop3<0x96>(x1, y3, kr), 8));
}
// One middle round of NB blocks: all the addresses, then all the look-ups, then the logic.  Left to
// itself the compiler issues the look-ups two at a time, each pair behind an s_waitcnt for the one before,
// so a wavefront never has more than two in flight.  Four wavefronts per SIMD fill the LDS pipe either
// way; the one or two per SIMD of the smaller workgroups (small batches) do not, and gain 7-8 % from
// the 16 NB look-ups of a round issued back to back (aes_rate; -DKY_AES_UNBATCHED for the other order).
template <int NB>
__device__ __forceinline__ void aes_round(const AesLds& L, uint32_t (&s)[NB][4], const uint32_t (&rk)[60], int r) {
#ifdef KY_AES_UNBATCHED
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const uint32_t t0 = aes_col(L, s[b][0], s[b][1], s[b][2], s[b][3], rk[4 * r]), t1 = aes_col(L, s[b][1], s[b][2], s[b][3], s[b][0], rk[4 * r + 1]);
    const uint32_t t2 = aes_col(L, s[b][2], s[b][3], s[b][0], s[b][1], rk[4 * r + 2]), t3 = aes_col(L, s[b][3], s[b][0], s[b][1], s[b][2], rk[4 * r + 3]);
    s[b][0] = t0; s[b][1] = t1; s[b][2] = t2; s[b][3] = t3;
  }
#else
  uint32_t x[NB][16];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      x[b][4 * c] = L.addr<0, 0>(s[b][c]);
      x[b][4 * c + 1] = L.addr<1, 0>(s[b][(c + 1) & 3]);
      x[b][4 * c + 2] = L.addr<2, 1>(s[b][(c + 2) & 3]);
      x[b][4 * c + 3] = L.addr<3, 1>(s[b][(c + 3) & 3]);
    }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int i = 0; i < 16; i++) x[b][i] = L.at(x[b][i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int c = 0; c < 4; c++)
      s[b][c] = bitop3<0x96>(x[b][4 * c], x[b][4 * c + 2], rotl32(bitop3<0x96>(x[b][4 * c + 1], x[b][4 * c + 3], rk[4 * r + c]), 8));
#endif
}
// one output column of the last round: S[a.b0] | S[b.b1] << 8 | S[c.b2] << 16 | S[d.b3] << 24, then ^ k.
// S sits in bytes 1 and 2 of a T0 entry and in bytes 0 and 3 of a T2 entry: three bit selects.
__device__ __forceinline__ uint32_t bitsel(uint32_t x, uint32_t y, uint32_t m) { return bitop3<0xE4>(x, y, m); }  // m ? x : y
__device__ __forceinline__ uint32_t aes_col_last(const AesLds& L, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t k) {
  const uint32_t e0 = L.look<0, 1>(a), e1 = L.look<1, 0>(b), e2 = L.look<2, 0>(c), e3 = L.look<3, 1>(d);
  return bitsel(bitsel(e0, e1, 0x000000FFu), bitsel(e2, e3, 0x00FF0000u), 0x0000FFFFu) ^ k;
}

// FIPS 197 §5.2 on little-endian packed words (byte 0 of a word = row 0).  On return the keys of
// the middle rounds (words 4..55) are rotated right by 8 bits, the form aes_col takes.
__device__ __forceinline__ void aes256_expand(const AesLds& L, const u64p (&key)[4], uint32_t (&rk)[60]) {
#pragma unroll
  for (int i = 0; i < 4; i++) { rk[2 * i] = key[i].lo; rk[2 * i + 1] = key[i].hi; }
  uint32_t rcon = 1;
#pragma unroll
  for (int i = 8; i < 60; i++) {
    uint32_t t = rk[i - 1];
    if (i % 8 == 0) {
      t = aes_subword(L, rotr32(t, 8)) ^ rcon;
      rcon = aes_xtime(rcon);
    } else if (i % 8 == 4) {
      t = aes_subword(L, t);
    }
    rk[i] = rk[i - 8] ^ t;
  }
#pragma unroll
  for (int i = 4; i < 56; i++) rk[i] = rotr32(rk[i], 8);
}

// Counter mode with the KEM's nonces (bytes 4..11 zero: ref/symmetric-aes.c:7-20):
// the input block is (n0, 0, 0, bswap(ctr)), so after the first AddRoundKey three
// of the four state words are the same for every block of a stream, and for
// ctr < 256 so are three bytes of the fourth.  Round 1 then has ONE look-up that
// depends on the counter (it lands in column 0), and round 2 four (one byte of
// that column per output column); the other 27 are folded once per stream into
// p[0] and q[] together with the round keys.  p[1..3] serve counters from 256 on,
// where the whole counter word varies and round 2 is a full round.
struct AesCtr {
  uint32_t p[4];
  uint32_t q[4];
};
__device__ __forceinline__ AesCtr aes256ctr_init(const AesLds& L, const uint32_t (&rk)[60], uint32_t n0) {
  const uint32_t s0 = n0 ^ rk[0], s1 = rk[1], s2 = rk[2];
  AesCtr c;
  // column = T0[a.b0] ^ T2[c.b2] ^ rotl8(T0[b.b1] ^ T2[d.b3] ^ kr), less the term that reads s3
  c.p[0] = bitop3<0x96>(L.look<0, 0>(s0), L.look<2, 1>(s2), rotl32(L.look<1, 0>(s1) ^ rk[4], 8));   // + rotl8 T2[s3.b3]
  c.p[1] = bitop3<0x96>(L.look<0, 0>(s1), rotl32(bitop3<0x96>(L.look<1, 0>(s2), L.look<3, 1>(s0), rk[5]), 8), 0u);  // + T2[s3.b2]
  c.p[2] = bitop3<0x96>(L.look<0, 0>(s2), L.look<2, 1>(s0), rotl32(L.look<3, 1>(s1) ^ rk[6], 8));   // + rotl8 T0[s3.b1]
  c.p[3] = bitop3<0x96>(L.look<2, 1>(s1), rotl32(bitop3<0x96>(L.look<1, 0>(s0), L.look<3, 1>(s2), rk[7]), 8), 0u);  // + T0[s3.b0]
  // ctr < 256: bytes 0..2 of s3 = those of rk[3], so words 1..3 after round 1 are known
  const uint32_t v = rk[3];
  const uint32_t u1 = c.p[1] ^ L.look<2, 1>(v), u2 = c.p[2] ^ rotl32(L.look<1, 0>(v), 8), u3 = c.p[3] ^ L.look<0, 0>(v);
  // round 2, less the term that reads word 0
  c.q[0] = L.look<2, 1>(u2) ^ rotl32(bitop3<0x96>(L.look<1, 0>(u1), L.look<3, 1>(u3), rk[8]), 8);          // + T0[u0.b0]
  c.q[1] = bitop3<0x96>(L.look<0, 0>(u1), L.look<2, 1>(u3), rotl32(L.look<1, 0>(u2) ^ rk[9], 8));          // + rotl8 T2[u0.b3]
  c.q[2] = L.look<0, 0>(u2) ^ rotl32(bitop3<0x96>(L.look<1, 0>(u3), L.look<3, 1>(u1), rk[10]), 8);         // + T2[u0.b2]
  c.q[3] = bitop3<0x96>(L.look<0, 0>(u3), L.look<2, 1>(u1), rotl32(L.look<3, 1>(u2) ^ rk[11], 8));         // + rotl8 T0[u0.b1]
  return c;
}
// state of block number ctr after round 2 (ctr is uniform over the wavefront wherever this is called)
__device__ __forceinline__ void aes256ctr_head(const AesLds& L, const uint32_t (&rk)[60], const AesCtr& c, uint32_t ctr, bool folded, uint32_t (&s)[4]) {
  const uint32_t v = bswap32(ctr) ^ rk[3];
  if (folded) {
    const uint32_t u0 = c.p[0] ^ rotl32(L.look<3, 1>(v), 8);
    s[0] = c.q[0] ^ L.look<0, 0>(u0);
    s[1] = c.q[1] ^ rotl32(L.look<3, 1>(u0), 8);
    s[2] = c.q[2] ^ L.look<2, 1>(u0);
    s[3] = c.q[3] ^ rotl32(L.look<1, 0>(u0), 8);
  } else {
    const uint32_t u0 = c.p[0] ^ rotl32(L.look<3, 1>(v), 8), u1 = c.p[1] ^ L.look<2, 1>(v), u2 = c.p[2] ^ rotl32(L.look<1, 0>(v), 8),
                   u3 = c.p[3] ^ L.look<0, 0>(v);
    s[0] = aes_col(L, u0, u1, u2, u3, rk[8]);
    s[1] = aes_col(L, u1, u2, u3, u0, rk[9]);
    s[2] = aes_col(L, u2, u3, u0, u1, rk[10]);
    s[3] = aes_col(L, u3, u0, u1, u2, rk[11]);
  }
}
// NB consecutive keystream blocks from number ctr on, advanced together round by round: the
// blocks are independent, so the look-ups of one are in flight while the logic of the others
// issues (aes_rate: +25 % over one block after the other at four wavefronts per SIMD).
template <int NB>
__device__ __forceinline__ void aes256ctr_blocks(const AesLds& L, const uint32_t (&rk)[60], const AesCtr& c, uint32_t ctr, uint32_t (&out)[NB][4]) {
  uint32_t s[NB][4];
  const bool folded = ctr < 256u - (NB - 1);  // one branch for the NB blocks
  if (folded) {
#pragma unroll
    for (int b = 0; b < NB; b++) aes256ctr_head(L, rk, c, ctr + b, true, s[b]);
  } else {
#pragma unroll
    for (int b = 0; b < NB; b++) aes256ctr_head(L, rk, c, ctr + b, false, s[b]);
  }
#pragma unroll
  for (int r = 3; r < 14; r++) aes_round<NB>(L, s, rk, r);
#pragma unroll
  for (int b = 0; b < NB; b++) {
    out[b][0] = aes_col_last(L, s[b][0], s[b][1], s[b][2], s[b][3], rk[56]);
    out[b][1] = aes_col_last(L, s[b][1], s[b][2], s[b][3], s[b][0], rk[57]);
    out[b][2] = aes_col_last(L, s[b][2], s[b][3], s[b][0], s[b][1], rk[58]);
    out[b][3] = aes_col_last(L, s[b][3], s[b][0], s[b][1], s[b][2], rk[59]);
  }
}

// keystream block number ctr
__device__ __forceinline__ void aes256ctr_block(const AesLds& L, const uint32_t (&rk)[60], const AesCtr& c, uint32_t ctr, uint32_t (&out)[4]) {
  uint32_t o[1][4];
  aes256ctr_blocks<1>(L, rk, c, ctr, o);
#pragma unroll
  for (int w = 0; w < 4; w++) out[w] = o[0][w];
}

}  // namespace ky
