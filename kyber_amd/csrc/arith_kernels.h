// arith_kernels.h -- the poly-layout kernels: 32 work-items per polynomial, 8 KEM
// instances per 256-thread workgroup, polynomials in LDS (see kem_kernels.h for
// the overall pipeline).  Compiled in their own translation unit
// (kyber_arith.hip) with the SLP vectorizer off: it packs the fp32 butterflies
// into v_pk_fma_f32 / v_pk_add_f32, which issue at half rate on gfx950 and need
// s_nop padding, and measured 5 % slower than the scalar fp32 stream.
//
// Reference call trees restated: ref/indcpa.c:205-329, ref/poly.c, ref/polyvec.c, ref/ntt.c.
#pragma once
#include "params.h"
#include "poly.h"
#include "fpoly.h"

namespace ky {

// ------------------------------------------------------------------ poly-layout helpers
struct Grp {
  int g, t;  // group in workgroup (0..7), work-item in group (0..31)
};
__device__ __forceinline__ Grp grp() { return Grp{(int)threadIdx.x >> 5, (int)threadIdx.x & 31}; }

__device__ __forceinline__ void ld_coef8(const int16_t* p, int t, int (&c)[8]) {
  unpack8(*reinterpret_cast<const uint4*>(p + 8 * t), c);
}

// The three arithmetic kernels below run on the fp32 engine of fpoly.h (plain
// residues, no Montgomery domain).  What they write -- pk, sk, ct, m' -- are
// canonical byte strings, identical to the reference's.

__device__ __forceinline__ void ld_coef8f(const int16_t* p, int t, float (&f)[8]) {
  int c[8];
  ld_coef8(p, t, c);
  i2f8(c, f);
}
__device__ __forceinline__ void ld_bytes12f(const uint8_t* p, int t, float (&f)[8]) {
  const uint32_t* s = reinterpret_cast<const uint32_t*>(p) + 3 * t;
  int c[8];
  frombytes12(s[0], s[1], s[2], c);
  i2f8(c, f);
}
__device__ __forceinline__ void st_bytes12(uint8_t* p, int t, const int (&c)[8]) {
  uint32_t w0, w1, w2;
  tobytes12(c, w0, w1, w2);
  uint32_t* o = reinterpret_cast<uint32_t*>(p) + 3 * t;
  o[0] = w0; o[1] = w1; o[2] = w2;
}

// ------------------------------------------------------------------ keypair arithmetic
// ref/indcpa.c:220-238: s,e <- CBD(PRF); s^,e^ = NTT; t^ = A s^ + e^;
// pk = tobytes(t^) || rho ; sk = tobytes(s^) (|| pk copy, ref/kem.c:28-29)
// Same shape as k_enc_arith: a 32-lane group owns one instance, s^ stays in
// registers as cached multiplicands for the K matrix rows, e^ waits in LDS.
template <int K>
struct FcGroups {  // polynomials that share one unreduced accumulation (at most three, fpoly.h)
  static constexpr int N = (K + 2) / 3;
  static constexpr int begin(int g) { return g * K / N; }
  static constexpr int end(int g) { return (g + 1) * K / N; }
};

template <class P>
__global__ void __launch_bounds__(256, P::K <= 3 ? 4 : 3) k_kp_arith(int n, const uint8_t* __restrict__ small,
                                                                      const int16_t* __restrict__ mat,
                                                                      const uint8_t* __restrict__ prf, uint8_t* __restrict__ pk,
                                                                      uint8_t* __restrict__ sk) {
  constexpr int K = P::K;
  using G = FcGroups<K>;
  __shared__ __attribute__((aligned(16))) float eh[G_INST][K][256];  // e^, read back CONTIG only
  __shared__ __attribute__((aligned(16))) float wk[8][256];
  __shared__ float ztab[128];
  const Grp w = grp();
  const int inst0 = blockIdx.x * G_INST;
  const int li = w.g;
  const int gi = min(inst0 + li, n - 1);
  const bool live = inst0 + li < n;
  f_twiddles_to_lds(ztab);

  const uint8_t* prf_i = prf + (size_t)gi * 2 * K * P::PRF_STRIDE;
  const int16_t* mat_i = mat + (size_t)gi * P::NMAT * 256;
  uint8_t* pk_i = pk + (size_t)gi * P::PKB;
  uint8_t* sk_i = sk + (size_t)gi * P::SKB;

  auto fetch_noise = [&](int v, uint32_t (&r)[3]) {
    const uint8_t* b = prf_i + (size_t)v * P::PRF_STRIDE;
    if constexpr (P::ETA1 == 2) {
      r[0] = *reinterpret_cast<const uint32_t*>(b + 4 * w.t);
    } else {
      const uint16_t* h = reinterpret_cast<const uint16_t*>(b + 6 * w.t);
      r[0] = h[0]; r[1] = h[1]; r[2] = h[2];
    }
  };
  auto fetch_row = [&](int i, uint4 (&a)[K]) {
#pragma unroll
    for (int j = 0; j < K; j++) a[j] = *reinterpret_cast<const uint4*>(mat_i + (i * K + j) * 256 + 8 * w.t);
  };

  FCache sc[K];
  float corr[G::N][8];
  uint4 nxt[K];
  {
    uint32_t nw[2 * K][3];
#pragma unroll
    for (int v = 0; v < 2 * K; v++) fetch_noise(v, nw[v]);
    fetch_row(0, nxt);
#pragma unroll
    for (int v = 0; v < 2 * K; v++) {
      int c[8];
      float f[8];
      if constexpr (P::ETA1 == 2) cbd2_word(nw[v][0], c);
      else cbd3_words(nw[v][0] | (nw[v][1] & 0xFFu) << 16, (nw[v][1] >> 8) | nw[v][2] << 8, c);
      i2f8(c, f);
      f_ntt_fwd<true>(wk[w.g], ztab, w.t, f);
      if (v < K) {
        f_cache_make(sc[v], f, ztab, w.t);
        f2canon8(f, c);  // sk = tobytes(s^)
        if (live) st_bytes12(sk_i + 384 * v, w.t, c);
      } else {
        f_put_plain(eh[li][v - K], w.t, f);
      }
    }
#pragma unroll
    for (int g = 0; g < G::N; g++) {
      float S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = G::begin(g); j < G::end(g); j++) f_cache_sum(S, sc[j]);
#pragma unroll
      for (int k = 0; k < 8; k++) corr[g][k] = f_corr(S[k]);
    }
  }
#pragma unroll 1
  for (int i = 0; i < K; i++) {
    uint4 cur[K];
#pragma unroll
    for (int j = 0; j < K; j++) cur[j] = nxt[j];
    if (i + 1 < K) fetch_row(i + 1, nxt);
    float acc[8], e[8];
#pragma unroll
    for (int g = 0; g < G::N; g++) {
#pragma unroll
      for (int k = 0; k < 8; k++) acc[k] = (g == 0) ? corr[0][k] : fred(acc[k]) + corr[g][k];
#pragma unroll
      for (int j = G::begin(g); j < G::end(g); j++) {
        float a[8];
        u16x8_to_f(cur[j], a);
        f_mac(acc, a, sc[j]);
      }
    }
    f_get_plain(eh[li][i], w.t, e);  // written by this work-item itself
    int c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = fcanon(fred(fred(acc[k]) + e[k]));  // + e^   ref/indcpa.c:234-235
    if (live) {
      st_bytes12(pk_i + 384 * i, w.t, c);
      st_bytes12(sk_i + P::PVB + 384 * i, w.t, c);
    }
  }
  // rho -> pk tail and its copy inside sk
  if (live && w.t < 8) {
    uint32_t r = reinterpret_cast<const uint32_t*>(small + (size_t)gi * SMALL)[w.t];
    reinterpret_cast<uint32_t*>(pk_i + P::PVB)[w.t] = r;
    reinterpret_cast<uint32_t*>(sk_i + 2 * P::PVB)[w.t] = r;
  }
}

// ------------------------------------------------------------------ encryption arithmetic
// ref/indcpa.c:268-295.  CMP = false: write the ciphertext.  CMP = true
// (decapsulation's re-encryption, ref/kem.c:114-116): compare with the given
// ciphertext instead and raise the instance's fail flag on any difference.
//
// One 32-lane group owns one instance from start to end: it transforms the K
// noise polynomials r_j once and keeps r^ in registers as cached multiplicands
// (fpoly.h FCache: converted, twisted and summed once instead of once per
// matrix row), then produces the K+1 result rows u_0..u_{K-1}, v one after the
// other.  Nothing is shared between groups before the ciphertext tile, so the
// only workgroup barrier is the one in front of the final copy-out.
template <class P, bool CMP>
__global__ void __launch_bounds__(256, P::K <= 3 ? 4 : 3) k_enc_arith(int n, const uint8_t* __restrict__ pk, size_t pk_stride,
                                                   uint8_t* __restrict__ small, const int16_t* __restrict__ mat,
                                                   size_t mat_stride,  // int16 per instance; 0 = one matrix for all
                                                   const uint8_t* __restrict__ prf, uint8_t* __restrict__ ct) {
  constexpr int K = P::K, NJ = 2 * K + 1;
  using G = FcGroups<K>;
  __shared__ __attribute__((aligned(16))) float wk[8][256];
  __shared__ __attribute__((aligned(16))) uint8_t ctb[G_INST][P::CTB];
  __shared__ float ztab[128];
  const Grp w = grp();
  const int inst0 = blockIdx.x * G_INST;
  const int li = w.g;                       // G_INST == number of groups
  const int gi = min(inst0 + li, n - 1);    // surplus groups of the last workgroup recompute the last instance
  f_twiddles_to_lds(ztab);

  const uint8_t* prf_i = prf + (size_t)gi * NJ * P::PRF_STRIDE;
  const int16_t* mat_i = mat + (size_t)gi * mat_stride;
  const uint8_t* pk_i = pk + (size_t)gi * pk_stride;

  // Global operands are fetched one step ahead of their use (the transforms'
  // dependent chains hide an HBM round trip that 3-4 waves per SIMD would not).
  struct Ops {
    uint4 a[K];           // matrix row entries (8 int16) or 12 key bytes in x,y,z
    uint32_t noise, msg;  // PRF word for CBD2; message byte (row K only)
  };
  auto fetch_row = [&](int i, Ops& o) {
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (i < K) {
        o.a[j] = *reinterpret_cast<const uint4*>(mat_i + (i * K + j) * 256 + 8 * w.t);
      } else {
        const uint32_t* sb = reinterpret_cast<const uint32_t*>(pk_i + 384 * j) + 3 * w.t;
        o.a[j] = make_uint4(sb[0], sb[1], sb[2], 0u);
      }
    }
    o.noise = *reinterpret_cast<const uint32_t*>(prf_i + (size_t)(K + i) * P::PRF_STRIDE + 4 * w.t);  // nonce K+i
    o.msg = (i == K) ? small[(size_t)gi * SMALL + w.t] : 0u;
  };
  auto fetch_noise1 = [&](int j, uint32_t (&r)[3]) {
    const uint8_t* b = prf_i + (size_t)j * P::PRF_STRIDE;
    if constexpr (P::ETA1 == 2) {
      r[0] = *reinterpret_cast<const uint32_t*>(b + 4 * w.t);
    } else {
      const uint16_t* h = reinterpret_cast<const uint16_t*>(b + 6 * w.t);
      r[0] = h[0]; r[1] = h[1]; r[2] = h[2];
    }
  };

  // r^_j = NTT(CBD_eta1(PRF(r, j))), cached
  FCache rc[K];
  float corr[G::N][8];
  Ops nxt;
  {
    uint32_t nw[K][3];
#pragma unroll
    for (int j = 0; j < K; j++) fetch_noise1(j, nw[j]);
    fetch_row(0, nxt);
#pragma unroll
    for (int j = 0; j < K; j++) {
      int c[8];
      float f[8];
      if constexpr (P::ETA1 == 2) cbd2_word(nw[j][0], c);
      else cbd3_words(nw[j][0] | (nw[j][1] & 0xFFu) << 16, (nw[j][1] >> 8) | nw[j][2] << 8, c);
      // r is scaled by 128^-1 here (|.| <= 78), once per polynomial, so that the K+1 inverse
      // transforms below need no final scaling: invntt is linear in r^
#pragma unroll
      for (int k = 0; k < 8; k++) f[k] = (float)c[k] * KY_NINV_PLAIN;
      f_ntt_fwd<true>(wk[w.g], ztab, w.t, f);
      f_cache_make(rc[j], f, ztab, w.t);
    }
#pragma unroll
    for (int g = 0; g < G::N; g++) {
      float S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = G::begin(g); j < G::end(g); j++) f_cache_sum(S, rc[j]);
#pragma unroll
      for (int k = 0; k < 8; k++) corr[g][k] = f_corr(S[k]);
    }
  }

  // rows 0..K-1: u_i = invntt(At_i . r^) + e1_i ; row K: v = invntt(t^ . r^) + e2 + msg
#pragma unroll 1
  for (int i = 0; i <= K; i++) {
    Ops& cur = nxt;  // consumed by the multiply-adds, then refilled for the next row while this row is transformed
    float acc[8];
    if (i < K) {
      // canonical matrix entries against the offset-corrected start value
#pragma unroll
      for (int g = 0; g < G::N; g++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = (g == 0) ? corr[0][k] : fred(acc[k]) + corr[g][k];
#pragma unroll
        for (int j = G::begin(g); j < G::end(g); j++) {
          float a[8];
          u16x8_to_f(cur.a[j], a);
          f_mac(acc, a, rc[j]);
        }
      }
    } else {
      // key bytes carry any 12-bit value (ref/poly.c:142-143): centre them first
#pragma unroll
      for (int g = 0; g < G::N; g++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = (g == 0) ? 0.0f : fred(acc[k]);
#pragma unroll
        for (int j = G::begin(g); j < G::end(g); j++) {
          int ai[8];
          float a[8];
          frombytes12(cur.a[j].x, cur.a[j].y, cur.a[j].z, ai);
#pragma unroll
          for (int k = 0; k < 8; k++) a[k] = fred((float)ai[k]);
          f_mac(acc, a, rc[j]);
        }
      }
    }
    const uint32_t noise = cur.noise, msg = cur.msg;
    if (i < K) fetch_row(i + 1, nxt);
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = fred(acc[k]);
    f_ntt_inv<false>(wk[w.g], ztab, w.t, acc);  // |acc| <= 3330
    int e[8];
    unsigned c[8];
    cbd2_word(noise, e);
    if (i < K) {
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = fcompress<P::DU>(acc[k] + (float)e[k]);  // |.| <= 3330 + 2
      pack_bits_contig<P::DU>(ctb[li] + P::UB * i, w.t, c);
    } else {
      int m[8];
      frommsg_contig(msg, m);
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = fcompress<P::DV>(acc[k] + (float)(e[k] + m[k]));  // |.| <= 3330 + 2 + 1665 = 4997 <= 5000 (fcompress)
      pack_bits_contig<P::DV>(ctb[li] + P::UB * K, w.t, c);
    }
  }
  if constexpr (CMP) {
    // ref/verify.c:16-25: OR-reduce of the XOR, no branch on the (secret-dependent) result.  Each group
    // compares the ciphertext it has just assembled (its own lanes wrote all of it: a wavefront-level
    // fence is enough) with the one it was given; the 32 lanes' findings meet in a ballot, whose half
    // for this group is non-zero exactly if some byte differs.
    KY_WSYNC();
    // Everything the comparison needs is derived again from the work-item id, through a copy the
    // compiler cannot trace back: carried from the top of the kernel, the lane number, the instance
    // number and the ciphertext offsets stayed live through the row loop and pushed the K = 3 kernel
    // five registers over the 128 of its occupancy bound (20 bytes of scratch per lane, written and
    // read once: 344 MB of extra HBM writes per 2^19 instances).
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int t2 = (int)(tid & 31u), l2 = (int)(tid >> 5);
    const int i2 = (int)blockIdx.x * G_INST + l2, g2 = min(i2, n - 1);
    const uint4* mine = reinterpret_cast<const uint4*>(&ctb[l2][0]);
    const uint4* given = reinterpret_cast<const uint4*>(ct + (size_t)g2 * P::CTB);
    uint32_t d = 0;
#pragma unroll
    for (int o = 0; o < (P::CTB / 16 + 31) / 32; o++) {
      const int b = min(t2 + 32 * o, P::CTB / 16 - 1);   // the last pass re-reads the last block instead of branching
      const uint4 x = mine[b], y = given[b];
      d |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w);
    }
    const unsigned long long differ = __ballot(d != 0);
    const uint32_t flag = (uint32_t)(differ >> (32 * (l2 & 1))) != 0u;
    if (t2 == 0 && i2 < n) *reinterpret_cast<uint32_t*>(small + (size_t)i2 * SMALL + 96) = flag;
  } else {
    __syncthreads();
    // the workgroup's ciphertexts are one contiguous run of global memory
    const int nvalid = min(G_INST, n - inst0);
    const uint4* src = reinterpret_cast<const uint4*>(&ctb[0][0]);
    uint4* dst = reinterpret_cast<uint4*>(ct + (size_t)inst0 * P::CTB);
    for (int o = threadIdx.x; o < nvalid * (P::CTB / 16); o += 256) dst[o] = src[o];
  }
}

// ------------------------------------------------------------------ decryption arithmetic
// ref/indcpa.c:318-328: m' = tomsg(v - invntt(s^ . NTT(decompress(u))))
// One 32-lane group per instance, as in k_enc_arith.  The ciphertext is staged
// in LDS with 16-byte loads (its 10/11-bit fields start at byte offsets that
// would otherwise cost ten single-byte global loads per work-item and
// polynomial) and read back with aligned 32-bit loads (decompress_contig_lds32:
// same-box A/B 0.495 -> 0.471 ms per 2^19 against 16-bit loads); s^ arrives as
// 12 key bytes per work-item and polynomial.
// (72 VGPRs for K = 3: seven wavefronts per SIMD; a bound of eight spills)
template <class P>
__global__ void __launch_bounds__(256) k_dec_arith(int n, const uint8_t* __restrict__ ct, const uint8_t* __restrict__ sk,
                                                   size_t sk_stride, uint8_t* __restrict__ small) {
  constexpr int K = P::K, CT16 = P::CTB / 16;
  using G = FcGroups<K>;
  __shared__ __attribute__((aligned(16))) uint8_t ctb[G_INST][P::CTB];
  __shared__ __attribute__((aligned(16))) float wk[8][256];
  __shared__ float ztab[128];
  const Grp w = grp();
  const int inst0 = blockIdx.x * G_INST;
  const int li = w.g;
  const int gi = min(inst0 + li, n - 1);
  const uint8_t* sk_i = sk + (size_t)gi * sk_stride;
  {
    const uint4* src = reinterpret_cast<const uint4*>(ct + (size_t)gi * P::CTB);
    uint4* dst = reinterpret_cast<uint4*>(ctb[li]);
#pragma unroll
    for (int o = 0; o < (CT16 + 31) / 32; o++)
      if (o * 32 + w.t < CT16) dst[o * 32 + w.t] = src[o * 32 + w.t];
  }
  uint32_t sb[K][3];
#pragma unroll
  for (int j = 0; j < K; j++) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(sk_i + 384 * j) + 3 * w.t;
    sb[j][0] = p[0]; sb[j][1] = p[1]; sb[j][2] = p[2];
  }
  f_twiddles_to_lds(ztab);  // ends with a workgroup barrier: the staged ciphertext is visible too

  float acc[8];
#pragma unroll
  for (int g = 0; g < G::N; g++) {
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = (g == 0) ? 0.0f : fred(acc[k]);
#pragma unroll
    for (int j = G::begin(g); j < G::end(g); j++) {
      int c[8];
      float u[8], s[8];
      decompress_contig_lds32<P::DU>(ctb[li] + P::UB * j, w.t, c);
      i2f8(c, u);
      f_ntt_fwd(wk[w.g], ztab, w.t, u);  // centred, |u| <= 1665
      frombytes12(sb[j][0], sb[j][1], sb[j][2], c);
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = fred((float)c[k]);  // key bytes carry any 12-bit value: centre them
      FCache sc;
      f_cache_make(sc, s, ztab, w.t);
      f_mac(acc, u, sc);  // six products of centred values per coefficient at most (fpoly.h)
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = fred(acc[k]);
  f_ntt_inv(wk[w.g], ztab, w.t, acc);
  int v[8];
  decompress_contig_lds32<P::DV>(ctb[li] + P::UB * K, w.t, v);
  unsigned byte = 0;  // poly_tomsg: bit k = round(2 x / q) & 1       ref/poly.c:180-194
#pragma unroll
  for (int k = 0; k < 8; k++) byte |= fcompress<1>((float)v[k] - acc[k]) << k;  // |.| <= 3328 + 1668
  if (inst0 + li < n) small[(size_t)gi * SMALL + w.t] = (uint8_t)byte;
}

// ------------------------------------------------------------------ standalone polynomial kernels
// (BASELINE.json config 2: batched NTT / basemul / inverse NTT on dense
// arrays of int16[256] polynomials; 8 polynomials per 256-thread workgroup)
// The kernels here are the int16 engine: invntt_tomont, basemul and their fusion with the
// Montgomery path's own representatives in (-q, q) -- montgomery_reduce returns the member of
// the residue class nearest to p / 2^16, which depends on the exact integer p, not on its
// residue, so only the reference's integer sequence reproduces them.  They are the debugging
// API (stage outputs that compare bit for bit with the reference's intermediates); the fast
// path of config 2 is the wide fp32 engine in wpoly.h, whose outputs are canonical, and which
// also runs the forward transform (poly_ntt's Barrett-centred output depends on the residue only).
//
// The transform kernels are grid-stride loops: a 32-lane group walks over
// polynomials pi, pi + stride, ... and issues the 16-byte load of its next
// polynomial before it transforms the current one.
__device__ __forceinline__ uint4 ld_poly16(const int16_t* base, long pi, int t) {
  return *reinterpret_cast<const uint4*>(base + pi * 256 + 8 * t);
}
__device__ __forceinline__ void st_poly16(int16_t* base, long pi, int t, const int (&c)[8]) { lds_put_contig(base + pi * 256, t, c); }

__global__ void __launch_bounds__(256) k_poly_invntt(long npoly, int16_t* __restrict__ r) {
  __shared__ __attribute__((aligned(16))) int16_t wk[8][256];
  __shared__ int ztab[128];
  const Grp w = grp();
  twiddles_to_lds(ztab);
  const long stride = (long)gridDim.x * 8;
  long pi = (long)blockIdx.x * 8 + w.g;
  uint4 nxt = make_uint4(0, 0, 0, 0);
  if (pi < npoly) nxt = ld_poly16(r, pi, w.t);
  for (; pi < npoly; pi += stride) {
    const uint4 cur = nxt;
    if (pi + stride < npoly) nxt = ld_poly16(r, pi + stride, w.t);
    int c[8];
    unpack8(cur, c);
    ntt_inv(wk[w.g], ztab, w.t, c);
    stride_to_contig(wk[w.g], w.t, c);
    st_poly16(r, pi, w.t, c);
    KY_GSYNC();
  }
}

// r = a o b (Montgomery domain), no reduction beyond basemul's own.  ref/poly.c:273-280
__global__ void __launch_bounds__(256) k_poly_basemul(long npoly, int16_t* r, const int16_t* a, const int16_t* b) {  // r may alias a or b
  __shared__ int ztab[128];
  const Grp w = grp();
  twiddles_to_lds(ztab);
  const long pi = (long)blockIdx.x * 8 + w.g;
  if (pi >= npoly) return;
  int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, x[8], y[8];
  ld_coef8(a + pi * 256, w.t, x);
  ld_coef8(b + pi * 256, w.t, y);
  basemul_acc(acc, x, y, ztab, w.t);
  lds_put_contig(r + pi * 256, w.t, acc);
}

// fused: r = invntt_tomont( poly_ntt(a) o poly_ntt(b) ) -- one HBM round trip
// per operand, everything else in registers/LDS.
__global__ void __launch_bounds__(256) k_poly_mul_fused(long npoly, int16_t* r, const int16_t* a, const int16_t* b) {  // r may alias a or b
  __shared__ __attribute__((aligned(16))) int16_t wk[8][256];
  __shared__ int ztab[128];
  const Grp w = grp();
  twiddles_to_lds(ztab);
  const long stride = (long)gridDim.x * 8;
  long pi = (long)blockIdx.x * 8 + w.g;
  uint4 na = make_uint4(0, 0, 0, 0), nb = na;
  if (pi < npoly) {
    na = ld_poly16(a, pi, w.t);
    nb = ld_poly16(b, pi, w.t);
  }
  for (; pi < npoly; pi += stride) {
    const uint4 ca = na, cb = nb;
    if (pi + stride < npoly) {
      na = ld_poly16(a, pi + stride, w.t);
      nb = ld_poly16(b, pi + stride, w.t);
    }
    int x[8], y[8], acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    KY_GSYNC();
    *reinterpret_cast<uint4*>(wk[w.g] + 8 * w.t) = ca;
    KY_GSYNC();
    ntt_fwd(wk[w.g], ztab, w.t, x);
    KY_GSYNC();
    *reinterpret_cast<uint4*>(wk[w.g] + 8 * w.t) = cb;
    KY_GSYNC();
    ntt_fwd(wk[w.g], ztab, w.t, y);
    basemul_acc(acc, x, y, ztab, w.t);
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = barrett(acc[k]);
    ntt_inv(wk[w.g], ztab, w.t, acc);
    stride_to_contig(wk[w.g], w.t, acc);
    st_poly16(r, pi, w.t, acc);
  }
}

// ------------------------------------------------------------------ single-function stages
// The codecs, the noise map and the accumulated base multiplication as kernels of their own, built
// from the very device functions the KEM kernels above use, so that a parity failure of the KEM
// can be localised (tests/test_gpu_stages.py).  32 work-items per polynomial, 8 polynomials per
// workgroup, coefficients 8t..8t+7 per work-item, like everything else in this file.

// ref/cbd.c:58-107 (poly_cbd_eta1 / eta2): buf = ETA*64 bytes per polynomial, at stride bstride
template <int ETA>
__global__ void __launch_bounds__(256) k_st_cbd(long npoly, int16_t* r, const uint8_t* buf, size_t bstride) {
  const Grp w = grp();
  const long pi = (long)blockIdx.x * 8 + w.g;
  if (pi >= npoly) return;
  int c[8];
  cbd_contig<ETA>(buf + (size_t)pi * bstride, w.t, c);
  lds_put_contig(r + pi * 256, w.t, c);
}
// the same on the PRF kernel's output layout: job (inst, nonce) at (inst * nnonce + nonce) * PRF_STRIDE,
// eta1 for nonce < n_eta1 and eta2 = 2 above (ref/poly.c:208-232 as ref/indcpa.c:223-226,279-283 call it)
template <class P>
__global__ void __launch_bounds__(256) k_st_noise(long njobs, int nnonce, int n_eta1, int16_t* r, const uint8_t* prf) {
  const Grp w = grp();
  const long pi = (long)blockIdx.x * 8 + w.g;
  if (pi >= njobs) return;
  const int nonce = (int)(pi % nnonce);
  int c[8];
  if (P::ETA1 == 3 && nonce < n_eta1) cbd_contig<3>(prf + (size_t)pi * P::PRF_STRIDE, w.t, c);
  else cbd_contig<2>(prf + (size_t)pi * P::PRF_STRIDE, w.t, c);
  lds_put_contig(r + pi * 256, w.t, c);
}
// ref/poly.c:111-126 / :138-145
__global__ void __launch_bounds__(256) k_st_tobytes(long npoly, uint8_t* out, const int16_t* a) {
  const Grp w = grp();
  const long pi = (long)blockIdx.x * 8 + w.g;
  if (pi >= npoly) return;
  int c[8];
  ld_coef8(a + pi * 256, w.t, c);
  uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)pi * 384) + 3 * w.t;
  tobytes12(c, o[0], o[1], o[2]);
}
__global__ void __launch_bounds__(256) k_st_frombytes(long npoly, int16_t* r, const uint8_t* in) {
  const Grp w = grp();
  const long pi = (long)blockIdx.x * 8 + w.g;
  if (pi >= npoly) return;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(in + (size_t)pi * 384) + 3 * w.t;
  int c[8];
  frombytes12(p[0], p[1], p[2], c);
  lds_put_contig(r + pi * 256, w.t, c);
}
// d-bit compression as the KEM kernels do it: one FMA per coefficient (fcompress), then the bit packer.
// D = 4, 5: ref/poly.c:18-58; D = 10, 11: ref/polyvec.c:15-64 (per polynomial); D = 1: poly_tomsg, ref/poly.c:180-194
template <int D>
__global__ void __launch_bounds__(256) k_st_compress(long npoly, uint8_t* out, const int16_t* a) {
  const Grp w = grp();
  const long pi = (long)blockIdx.x * 8 + w.g;
  if (pi >= npoly) return;
  int c[8];
  ld_coef8(a + pi * 256, w.t, c);
  unsigned v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = fcompress<D>((float)c[k]);
  if constexpr (D == 1) {
    unsigned b = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) b |= v[k] << k;
    out[(size_t)pi * 32 + w.t] = (uint8_t)b;
  } else {
    pack_bits_contig<D>(out + (size_t)pi * 32 * D, w.t, v);
  }
}
// ref/poly.c:70-100, ref/polyvec.c:76-115; D = 1: poly_frommsg, ref/poly.c:155-170
template <int D>
__global__ void __launch_bounds__(256) k_st_decompress(long npoly, int16_t* r, const uint8_t* in) {
  const Grp w = grp();
  const long pi = (long)blockIdx.x * 8 + w.g;
  if (pi >= npoly) return;
  int c[8];
  if constexpr (D == 1) frommsg_contig(in[(size_t)pi * 32 + w.t], c);
  else decompress_contig<D>(in + (size_t)pi * 32 * D, w.t, c);
  lds_put_contig(r + pi * 256, w.t, c);
}
// sum_j a_j o b_j over K polynomial pairs on the fp32 engine, exactly as k_enc_arith accumulates a
// matrix row against the cached r^: a canonical in [0, q) (matrix entries), b centred (|b| <= 1665,
// an NTT output), result the centred residue of the PLAIN product -- the reference's
// polyvec_basemul_acc_montgomery (ref/polyvec.c:189-201) carries a factor 2^-16 on top, so the test
// compares r * 2^-16 mod q.
template <int K>
__global__ void __launch_bounds__(256) k_st_basemul_acc(long nvec, int16_t* r, const int16_t* a, const int16_t* b) {
  __shared__ float ztab[128];
  const Grp w = grp();
  f_twiddles_to_lds(ztab);
  const long vi = (long)blockIdx.x * 8 + w.g;
  if (vi >= nvec) return;
  using G = FcGroups<K>;
  FCache bc[K];
  float corr[G::N][8], acc[8];
#pragma unroll
  for (int g = 0; g < G::N; g++) {
    float S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = G::begin(g); j < G::end(g); j++) {
      int c[8];
      float f[8];
      ld_coef8(b + (vi * K + j) * 256, w.t, c);
      i2f8(c, f);
      f_cache_make(bc[j], f, ztab, w.t);
      f_cache_sum(S, bc[j]);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) corr[g][k] = f_corr(S[k]);
  }
#pragma unroll
  for (int g = 0; g < G::N; g++) {
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = (g == 0) ? corr[0][k] : fred(acc[k]) + corr[g][k];
#pragma unroll
    for (int j = G::begin(g); j < G::end(g); j++) {
      float f[8];
      u16x8_to_f(*reinterpret_cast<const uint4*>(a + (vi * K + j) * 256 + 8 * w.t), f);
      f_mac(acc, f, bc[j]);
    }
  }
  int c[8];
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = (int)fred(acc[k]);
  lds_put_contig(r + vi * 256, w.t, c);
}

}  // namespace ky
