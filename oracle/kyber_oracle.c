/*
 * kyber_oracle.c -- CPU restatement of round-3 Kyber (see kyber_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into, loaded by or called from the
 * product library.
 *
 * Written from the algorithm, not from the reference's text: one run-time
 * parameter block instead of compile-time sets, table-free Keccak (rotation
 * offsets and round constants derived on first use), a byte-oriented AES, a
 * single generic d-bit compressor.  Each function names the reference lines it
 * restates so parity can be audited.
 */
#include "kyber_oracle.h"
#include "sha2_consts.h"
#include <string.h>

/* ------------------------------------------------------------------ */
/* parameters                                                          */
/* ------------------------------------------------------------------ */
int ko_params_init(ko_params *p, int k, int nineties)
{
  if (k < 2 || k > 4) return -1;
  p->k = k;
  p->eta1 = (k == 2) ? 3 : 2;
  p->du = (k == 4) ? 11 : 10;
  p->dv = (k == 4) ? 5 : 4;
  p->nineties = nineties ? 1 : 0;
  return 0;
}
static size_t polyvec_bytes(const ko_params *p) { return (size_t)p->k * 384; }
static size_t ucomp_bytes(const ko_params *p) { return (size_t)p->k * 32 * p->du; }
static size_t vcomp_bytes(const ko_params *p) { return (size_t)32 * p->dv; }
size_t ko_pk_bytes(const ko_params *p) { return polyvec_bytes(p) + 32; }
size_t ko_sk_bytes(const ko_params *p) { return 2 * polyvec_bytes(p) + 32 + 64; }
size_t ko_ct_bytes(const ko_params *p) { return ucomp_bytes(p) + vcomp_bytes(p); }

/* ------------------------------------------------------------------ */
/* Keccak-f[1600] and the four FIPS 202 functions Kyber uses            */
/* ------------------------------------------------------------------ */
static inline uint64_t rol64(uint64_t x, unsigned n) { return n ? (x << n) | (x >> (64 - n)) : x; }

static uint64_t kc_rc[24];
static unsigned kc_rot[24], kc_dst[24];
static int kc_ready;

static void kc_setup(void)
{
  unsigned x = 1, y = 0, t, lfsr = 1;
  for (t = 0; t < 24; t++) { /* rho/pi walk, FIPS 202 §3.2.2-3 */
    unsigned ny = (2 * x + 3 * y) % 5;
    x = y; y = ny;
    kc_rot[t] = ((t + 1) * (t + 2) / 2) % 64;
    kc_dst[t] = x + 5 * y;
  }
  for (t = 0; t < 24; t++) { /* iota constants, FIPS 202 §3.2.5 */
    uint64_t c = 0;
    for (unsigned j = 0; j < 7; j++) {
      unsigned bit = lfsr & 1;
      lfsr = (lfsr & 0x80) ? ((lfsr << 1) ^ 0x71) & 0xFF : (lfsr << 1);
      if (bit) c ^= 1ULL << ((1u << j) - 1);
    }
    kc_rc[t] = c;
  }
  kc_ready = 1;
}

/* ref/fips202.c:82-344 (there fully unrolled, two rounds per iteration) */
void ko_keccak_f1600(uint64_t s[25])
{
  if (!kc_ready) kc_setup();
  for (int rnd = 0; rnd < 24; rnd++) {
    uint64_t c[5], cur, nxt;
    for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
    for (int x = 0; x < 5; x++) {
      uint64_t d = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
      for (int y = 0; y < 25; y += 5) s[x + y] ^= d;
    }
    cur = s[1];
    for (int t = 0; t < 24; t++) {
      nxt = s[kc_dst[t]];
      s[kc_dst[t]] = rol64(cur, kc_rot[t]);
      cur = nxt;
    }
    for (int y = 0; y < 25; y += 5) {
      uint64_t r[5];
      for (int x = 0; x < 5; x++) r[x] = s[y + x];
      for (int x = 0; x < 5; x++) s[y + x] = r[x] ^ (~r[(x + 1) % 5] & r[(x + 2) % 5]);
    }
    s[0] ^= kc_rc[rnd];
  }
}

typedef struct { uint64_t s[25]; unsigned rate, pos; } sponge;

/* ref/fips202.c:461-485 keccak_absorb_once (+ pos bookkeeping of :580-584) */
static void sponge_absorb(sponge *sp, unsigned rate, const uint8_t *in, size_t inlen, uint8_t dom)
{
  memset(sp->s, 0, sizeof sp->s);
  sp->rate = rate;
  unsigned i = 0;
  while (inlen--) {
    sp->s[i >> 3] ^= (uint64_t)*in++ << (8 * (i & 7));
    if (++i == rate) { ko_keccak_f1600(sp->s); i = 0; }
  }
  sp->s[i >> 3] ^= (uint64_t)dom << (8 * (i & 7));
  sp->s[(rate - 1) >> 3] ^= 1ULL << 63;
  sp->pos = rate; /* nothing squeezed yet: next byte needs a permutation */
}

/* ref/fips202.c:426-446 keccak_squeeze / :500-514 keccak_squeezeblocks */
static void sponge_squeeze(sponge *sp, uint8_t *out, size_t outlen)
{
  while (outlen--) {
    if (sp->pos == sp->rate) { ko_keccak_f1600(sp->s); sp->pos = 0; }
    *out++ = (uint8_t)(sp->s[sp->pos >> 3] >> (8 * (sp->pos & 7)));
    sp->pos++;
  }
}

void ko_shake128(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen)
{ sponge sp; sponge_absorb(&sp, 168, in, inlen, 0x1F); sponge_squeeze(&sp, out, outlen); }
void ko_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen)
{ sponge sp; sponge_absorb(&sp, 136, in, inlen, 0x1F); sponge_squeeze(&sp, out, outlen); }
void ko_sha3_256(uint8_t h[32], const uint8_t *in, size_t inlen)
{ sponge sp; sponge_absorb(&sp, 136, in, inlen, 0x06); sponge_squeeze(&sp, h, 32); }
void ko_sha3_512(uint8_t h[64], const uint8_t *in, size_t inlen)
{ sponge sp; sponge_absorb(&sp, 72, in, inlen, 0x06); sponge_squeeze(&sp, h, 64); }

/* ------------------------------------------------------------------ */
/* SHA-256 / SHA-512 (FIPS 180-4); reference: ref/sha256.c, ref/sha512.c */
/* ------------------------------------------------------------------ */
static inline uint32_t ror32(uint32_t x, unsigned n) { return (x >> n) | (x << (32 - n)); }
static inline uint64_t ror64(uint64_t x, unsigned n) { return (x >> n) | (x << (64 - n)); }

static void sha256_block(uint32_t h[8], const uint8_t *b)
{
  uint32_t w[64], v[8];
  for (int i = 0; i < 16; i++)
    w[i] = (uint32_t)b[4 * i] << 24 | (uint32_t)b[4 * i + 1] << 16 | (uint32_t)b[4 * i + 2] << 8 | b[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  memcpy(v, h, sizeof v);
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = ror32(v[4], 6) ^ ror32(v[4], 11) ^ ror32(v[4], 25);
    uint32_t ch = (v[4] & v[5]) ^ (~v[4] & v[6]);
    uint32_t t1 = v[7] + S1 + ch + KO_SHA256_K[i] + w[i];
    uint32_t S0 = ror32(v[0], 2) ^ ror32(v[0], 13) ^ ror32(v[0], 22);
    uint32_t mj = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]);
    memmove(v + 1, v, 7 * sizeof v[0]);
    v[4] += t1;
    v[0] = t1 + S0 + mj;
  }
  for (int i = 0; i < 8; i++) h[i] += v[i];
}

void ko_sha256(uint8_t out[32], const uint8_t *in, size_t inlen)
{
  uint32_t h[8];
  uint8_t tail[128];
  uint64_t bits = (uint64_t)inlen * 8;
  memcpy(h, KO_SHA256_H0, sizeof h);
  for (; inlen >= 64; in += 64, inlen -= 64) sha256_block(h, in);
  size_t tl = (inlen < 56) ? 64 : 128;
  memset(tail, 0, sizeof tail);
  memcpy(tail, in, inlen);
  tail[inlen] = 0x80;
  for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
  for (size_t o = 0; o < tl; o += 64) sha256_block(h, tail + o);
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(h[i] >> (24 - 8 * j));
}

static void sha512_block(uint64_t h[8], const uint8_t *b)
{
  uint64_t w[80], v[8];
  for (int i = 0; i < 16; i++) {
    uint64_t x = 0;
    for (int j = 0; j < 8; j++) x = x << 8 | b[8 * i + j];
    w[i] = x;
  }
  for (int i = 16; i < 80; i++) {
    uint64_t s0 = ror64(w[i - 15], 1) ^ ror64(w[i - 15], 8) ^ (w[i - 15] >> 7);
    uint64_t s1 = ror64(w[i - 2], 19) ^ ror64(w[i - 2], 61) ^ (w[i - 2] >> 6);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  memcpy(v, h, sizeof v);
  for (int i = 0; i < 80; i++) {
    uint64_t S1 = ror64(v[4], 14) ^ ror64(v[4], 18) ^ ror64(v[4], 41);
    uint64_t ch = (v[4] & v[5]) ^ (~v[4] & v[6]);
    uint64_t t1 = v[7] + S1 + ch + KO_SHA512_K[i] + w[i];
    uint64_t S0 = ror64(v[0], 28) ^ ror64(v[0], 34) ^ ror64(v[0], 39);
    uint64_t mj = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]);
    memmove(v + 1, v, 7 * sizeof v[0]);
    v[4] += t1;
    v[0] = t1 + S0 + mj;
  }
  for (int i = 0; i < 8; i++) h[i] += v[i];
}

void ko_sha512(uint8_t out[64], const uint8_t *in, size_t inlen)
{
  uint64_t h[8];
  uint8_t tail[256];
  uint64_t bits = (uint64_t)inlen * 8;
  memcpy(h, KO_SHA512_H0, sizeof h);
  for (; inlen >= 128; in += 128, inlen -= 128) sha512_block(h, in);
  size_t tl = (inlen < 112) ? 128 : 256;
  memset(tail, 0, sizeof tail);
  memcpy(tail, in, inlen);
  tail[inlen] = 0x80;
  for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
  for (size_t o = 0; o < tl; o += 128) sha512_block(h, tail + o);
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(h[i] >> (56 - 8 * j));
}

/* ------------------------------------------------------------------ */
/* AES-256-CTR (FIPS 197).  The reference's bitsliced engine              */
/* (ref/aes256ctr.c:339-520) computes the same cipher; the counter layout */
/* restated here is ref/aes256ctr.c:530-543, :565-577.                    */
/* ------------------------------------------------------------------ */
static uint8_t aes_sbox[256];
static int aes_ready;

static uint8_t gf_mul(uint8_t a, uint8_t b)
{
  uint8_t r = 0;
  while (b) {
    if (b & 1) r ^= a;
    a = (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1B : 0));
    b >>= 1;
  }
  return r;
}

static void aes_setup(void)
{
  for (int a = 0; a < 256; a++) {
    uint8_t inv = 0;
    for (int b = 1; b < 256 && a; b++)
      if (gf_mul((uint8_t)a, (uint8_t)b) == 1) { inv = (uint8_t)b; break; }
    uint8_t y = inv;
    for (int s = 1; s <= 4; s++) y ^= (uint8_t)((inv << s) | (inv >> (8 - s)));
    aes_sbox[a] = y ^ 0x63;
  }
  aes_ready = 1;
}

static void aes256_expand(uint8_t rk[240], const uint8_t key[32])
{
  uint8_t rcon = 1;
  memcpy(rk, key, 32);
  for (int i = 8; i < 60; i++) {
    uint8_t t[4];
    memcpy(t, rk + 4 * (i - 1), 4);
    if (i % 8 == 0) {
      uint8_t t0 = t[0];
      t[0] = aes_sbox[t[1]] ^ rcon; t[1] = aes_sbox[t[2]]; t[2] = aes_sbox[t[3]]; t[3] = aes_sbox[t0];
      rcon = gf_mul(rcon, 2);
    } else if (i % 8 == 4) {
      for (int j = 0; j < 4; j++) t[j] = aes_sbox[t[j]];
    }
    for (int j = 0; j < 4; j++) rk[4 * i + j] = rk[4 * (i - 8) + j] ^ t[j];
  }
}

static void aes256_encrypt(uint8_t out[16], const uint8_t in[16], const uint8_t rk[240])
{
  uint8_t s[16], t[16];
  for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[i];
  for (int rnd = 1; rnd <= 14; rnd++) {
    for (int c = 0; c < 4; c++)          /* SubBytes + ShiftRows */
      for (int r = 0; r < 4; r++) t[4 * c + r] = aes_sbox[s[4 * ((c + r) & 3) + r]];
    if (rnd < 14) {
      for (int c = 0; c < 4; c++) {      /* MixColumns */
        uint8_t *a = t + 4 * c, x = a[0] ^ a[1] ^ a[2] ^ a[3];
        for (int r = 0; r < 4; r++) s[4 * c + r] = a[r] ^ x ^ gf_mul(a[r] ^ a[(r + 1) & 3], 2);
      }
    } else {
      memcpy(s, t, 16);
    }
    for (int i = 0; i < 16; i++) s[i] ^= rk[16 * rnd + i];
  }
  memcpy(out, s, 16);
}

typedef struct { uint8_t rk[240]; uint8_t nonce[12]; uint32_t ctr; } aesctr;

static void aesctr_init(aesctr *c, const uint8_t key[32], const uint8_t nonce[12], uint32_t ctr0)
{
  if (!aes_ready) aes_setup();
  aes256_expand(c->rk, key);
  memcpy(c->nonce, nonce, 12);
  c->ctr = ctr0;
}

static void aesctr_block(aesctr *c, uint8_t out[16])
{
  uint8_t blk[16];
  memcpy(blk, c->nonce, 12);
  blk[12] = (uint8_t)(c->ctr >> 24); blk[13] = (uint8_t)(c->ctr >> 16);
  blk[14] = (uint8_t)(c->ctr >> 8);  blk[15] = (uint8_t)c->ctr;
  c->ctr++;
  aes256_encrypt(out, blk, c->rk);
}

void ko_aes256ctr(uint8_t *out, size_t outlen, const uint8_t key[32], const uint8_t nonce[12], uint32_t ctr0)
{
  aesctr c;
  uint8_t blk[16];
  aesctr_init(&c, key, nonce, ctr0);
  while (outlen) {
    size_t n = outlen < 16 ? outlen : 16;
    aesctr_block(&c, blk);
    memcpy(out, blk, n);
    out += n; outlen -= n;
  }
}

/* ------------------------------------------------------------------ */
/* the symmetric switch, ref/symmetric.h:27-32 and :51-56               */
/* ------------------------------------------------------------------ */
static void hash_h(const ko_params *p, uint8_t out[32], const uint8_t *in, size_t n)
{ if (p->nineties) ko_sha256(out, in, n); else ko_sha3_256(out, in, n); }
static void hash_g(const ko_params *p, uint8_t out[64], const uint8_t *in, size_t n)
{ if (p->nineties) ko_sha512(out, in, n); else ko_sha3_512(out, in, n); }
static void kdf_n(const ko_params *p, uint8_t out[32], const uint8_t *in, size_t n)
{ if (p->nineties) ko_sha256(out, in, n); else ko_shake256(out, 32, in, n); }
static void kdf(const ko_params *p, uint8_t out[32], const uint8_t in[64]) { kdf_n(p, out, in, 64); }

/* ref/symmetric-shake.c:43-51 and ref/symmetric-aes.c:15-20 */
static void prf(const ko_params *p, uint8_t *out, size_t outlen, const uint8_t key[32], uint8_t nonce)
{
  if (p->nineties) {
    uint8_t iv[12] = {0};
    iv[0] = nonce;
    ko_aes256ctr(out, outlen, key, iv, 0);
  } else {
    uint8_t ext[33];
    memcpy(ext, key, 32);
    ext[32] = nonce;
    ko_shake256(out, outlen, ext, 33);
  }
}

/* The matrix XOF seen as one byte stream: SHAKE128(rho||x||y)
 * (ref/symmetric-shake.c:18-30) or AES-256-CTR(key=rho, iv={x,y,0..})
 * (ref/symmetric-aes.c:7-13).  `bytes` counts what has been drawn so callers
 * can report the number of XOF blocks the reference would have squeezed. */
typedef struct { int nineties; sponge sp; aesctr ac; uint8_t blk[16]; unsigned blkpos; size_t bytes; } xof;

static void xof_start(xof *x, const ko_params *p, const uint8_t seed[32], uint8_t a, uint8_t b)
{
  x->nineties = p->nineties;
  x->bytes = 0;
  if (p->nineties) {
    uint8_t iv[12] = {0};
    iv[0] = a; iv[1] = b;
    aesctr_init(&x->ac, seed, iv, 0);
    x->blkpos = 16;
  } else {
    uint8_t ext[34];
    memcpy(ext, seed, 32);
    ext[32] = a; ext[33] = b;
    sponge_absorb(&x->sp, 168, ext, 34, 0x1F);
  }
}

static uint8_t xof_byte(xof *x)
{
  uint8_t v;
  x->bytes++;
  if (!x->nineties) { sponge_squeeze(&x->sp, &v, 1); return v; }
  if (x->blkpos == 16) { aesctr_block(&x->ac, x->blk); x->blkpos = 0; }
  return x->blk[x->blkpos++];
}

/* ------------------------------------------------------------------ */
/* modular arithmetic                                                   */
/* ------------------------------------------------------------------ */
#define QINV (-3327) /* q^-1 mod 2^16, ref/reduce.h:8 */

int16_t ko_montgomery_reduce(int32_t a) /* ref/reduce.c:16-23 */
{
  int16_t m = (int16_t)((int16_t)a * (int16_t)QINV);
  return (int16_t)((a - (int32_t)m * KO_Q) >> 16);
}

int16_t ko_barrett_reduce(int16_t a) /* ref/reduce.c:35-42 */
{
  const int32_t v = ((1 << 26) + KO_Q / 2) / KO_Q; /* 20159 */
  int16_t t = (int16_t)((v * a + (1 << 25)) >> 26);
  return (int16_t)(a - t * KO_Q);
}

static int16_t fqmul(int16_t a, int16_t b) { return ko_montgomery_reduce((int32_t)a * b); }

static int16_t zt[128];
static int zt_ready;

/* same procedure as the generator kept in a comment at ref/ntt.c:6-36:
 * zeta_i = 2^16 * 17^brv7(i) mod q, centred into [-(q-1)/2, (q-1)/2] */
const int16_t *ko_zetas(void)
{
  if (!zt_ready) {
    int32_t pw[128];
    pw[0] = 1;
    for (int i = 1; i < 128; i++) pw[i] = pw[i - 1] * 17 % KO_Q;
    for (int i = 0; i < 128; i++) {
      int r = 0;
      for (int b = 0; b < 7; b++) r |= ((i >> b) & 1) << (6 - b);
      int32_t z = (int32_t)(((int64_t)pw[r] << 16) % KO_Q);
      if (z > KO_Q / 2) z -= KO_Q;
      zt[i] = (int16_t)z;
    }
    zt_ready = 1;
  }
  return zt;
}

void ko_ntt(int16_t r[256]) /* ref/ntt.c:80-95 */
{
  const int16_t *z = ko_zetas();
  int k = 1;
  for (int len = 128; len >= 2; len >>= 1)
    for (int start = 0; start < 256; start += 2 * len) {
      int16_t zeta = z[k++];
      for (int j = start; j < start + len; j++) {
        int16_t t = fqmul(zeta, r[j + len]);
        r[j + len] = (int16_t)(r[j] - t);
        r[j] = (int16_t)(r[j] + t);
      }
    }
}

void ko_poly_ntt(int16_t r[256]) /* ref/poly.c:244-248 */
{
  ko_ntt(r);
  for (int i = 0; i < 256; i++) r[i] = ko_barrett_reduce(r[i]);
}

void ko_invntt_tomont(int16_t r[256]) /* ref/ntt.c:106-126 */
{
  const int16_t *z = ko_zetas();
  int k = 127;
  for (int len = 2; len <= 128; len <<= 1)
    for (int start = 0; start < 256; start += 2 * len) {
      int16_t zeta = z[k--];
      for (int j = start; j < start + len; j++) {
        int16_t t = r[j];
        r[j] = ko_barrett_reduce((int16_t)(t + r[j + len]));
        r[j + len] = fqmul(zeta, (int16_t)(r[j + len] - t));
      }
    }
  for (int j = 0; j < 256; j++) r[j] = fqmul(r[j], 1441); /* 2^32/128 mod q */
}

/* ref/ntt.c:139-146 applied as in ref/poly.c:273-280 */
void ko_poly_basemul(int16_t r[256], const int16_t a[256], const int16_t b[256])
{
  const int16_t *z = ko_zetas();
  for (int i = 0; i < 128; i++) {
    int16_t zeta = (i & 1) ? (int16_t)-z[64 + i / 2] : z[64 + i / 2];
    const int16_t *x = a + 2 * i, *y = b + 2 * i;
    int16_t hi = fqmul(fqmul(x[1], y[1]), zeta);
    r[2 * i] = (int16_t)(hi + fqmul(x[0], y[0]));
    r[2 * i + 1] = (int16_t)(fqmul(x[0], y[1]) + fqmul(x[1], y[0]));
  }
}

void ko_polyvec_basemul_acc(int k, int16_t r[256], const int16_t *a, const int16_t *b) /* ref/polyvec.c:189-201 */
{
  int16_t t[256];
  ko_poly_basemul(r, a, b);
  for (int i = 1; i < k; i++) {
    ko_poly_basemul(t, a + 256 * i, b + 256 * i);
    for (int j = 0; j < 256; j++) r[j] = (int16_t)(r[j] + t[j]);
  }
  for (int j = 0; j < 256; j++) r[j] = ko_barrett_reduce(r[j]);
}

void ko_poly_tomont(int16_t r[256]) /* ref/poly.c:290-296 */
{
  for (int i = 0; i < 256; i++) r[i] = ko_montgomery_reduce((int32_t)r[i] * 1353); /* 2^32 mod q */
}

/* ------------------------------------------------------------------ */
/* sampling                                                             */
/* ------------------------------------------------------------------ */
/* ref/cbd.c:58-75 (eta=2) and :89-107 (eta=3): coefficient i is
 * popcount(bits [2*eta*i, +eta)) - popcount(bits [2*eta*i+eta, +eta)) */
void ko_cbd(int eta, int16_t r[256], const uint8_t *buf)
{
  for (int i = 0; i < 256; i++) {
    int a = 0, b = 0, base = 2 * eta * i;
    for (int j = 0; j < eta; j++) {
      a += (buf[(base + j) >> 3] >> ((base + j) & 7)) & 1;
      b += (buf[(base + eta + j) >> 3] >> ((base + eta + j) & 7)) & 1;
    }
    r[i] = (int16_t)(a - b);
  }
}

void ko_getnoise(const ko_params *p, int eta, int16_t r[256], const uint8_t seed[32], uint8_t nonce)
{
  uint8_t buf[192];
  prf(p, buf, (size_t)eta * 64, seed, nonce); /* ref/poly.c:208-232 */
  ko_cbd(eta, r, buf);
}

/* ref/indcpa.c:124-145 + :164-192.  The reference refills a buffer block by
 * block and carries buflen%3 bytes over; that is the same as reading the XOF
 * as one stream three bytes at a time until 256 candidates are accepted. */
void ko_gen_matrix(const ko_params *p, int16_t *a, const uint8_t seed[32], int transposed, int *nblocks_out)
{
  const size_t first = p->nineties ? 8 * 64 : 3 * 168, blk = p->nineties ? 64 : 168;
  for (int i = 0; i < p->k; i++)
    for (int j = 0; j < p->k; j++) {
      xof x;
      int16_t *r = a + 256 * (i * p->k + j);
      int ctr = 0;
      if (transposed) xof_start(&x, p, seed, (uint8_t)i, (uint8_t)j);
      else xof_start(&x, p, seed, (uint8_t)j, (uint8_t)i);
      while (ctr < 256) {
        unsigned b0 = xof_byte(&x), b1 = xof_byte(&x), b2 = xof_byte(&x);
        unsigned v0 = (b0 | b1 << 8) & 0xFFF, v1 = (b1 >> 4 | b2 << 4) & 0xFFF;
        if (v0 < KO_Q) r[ctr++] = (int16_t)v0;
        if (ctr < 256 && v1 < KO_Q) r[ctr++] = (int16_t)v1;
      }
      if (nblocks_out) {
        size_t nb = first / blk;
        if (x.bytes > first) nb += (x.bytes - first + blk - 1) / blk;
        nblocks_out[i * p->k + j] = (int)nb;
      }
    }
}

/* ------------------------------------------------------------------ */
/* byte codecs                                                          */
/* ------------------------------------------------------------------ */
static inline uint16_t canon(int16_t a) { return (uint16_t)(a + ((a >> 15) & KO_Q)); } /* ref/poly.c:118-121 */

void ko_poly_tobytes(uint8_t r[384], const int16_t a[256])
{
  for (int i = 0; i < 128; i++) {
    uint16_t t0 = canon(a[2 * i]), t1 = canon(a[2 * i + 1]);
    r[3 * i] = (uint8_t)t0;
    r[3 * i + 1] = (uint8_t)(t0 >> 8 | t1 << 4);
    r[3 * i + 2] = (uint8_t)(t1 >> 4);
  }
}

void ko_poly_frombytes(int16_t r[256], const uint8_t a[384])
{
  for (int i = 0; i < 128; i++) {
    r[2 * i] = (int16_t)((a[3 * i] | (uint16_t)a[3 * i + 1] << 8) & 0xFFF);
    r[2 * i + 1] = (int16_t)((a[3 * i + 1] >> 4 | (uint16_t)a[3 * i + 2] << 4) & 0xFFF);
  }
}

/* One d-bit compressor for d in {4,5,10,11}: coefficient i occupies bits
 * [d*i, d*i+d) of the little-endian output.  The four hand-packed layouts of
 * ref/poly.c:24-57 and ref/polyvec.c:19-63 are exactly this bit order. */
void ko_poly_compress(int d, uint8_t *r, const int16_t a[256])
{
  memset(r, 0, (size_t)32 * d);
  for (int i = 0; i < 256; i++) {
    uint32_t t = ((((uint32_t)canon(a[i]) << d) + KO_Q / 2) / KO_Q) & ((1u << d) - 1);
    for (int b = 0; b < d; b++)
      r[(d * i + b) >> 3] |= (uint8_t)(((t >> b) & 1) << ((d * i + b) & 7));
  }
}

void ko_poly_decompress(int d, int16_t r[256], const uint8_t *a)
{
  for (int i = 0; i < 256; i++) {
    uint32_t t = 0;
    for (int b = 0; b < d; b++) t |= (uint32_t)((a[(d * i + b) >> 3] >> ((d * i + b) & 7)) & 1) << b;
    r[i] = (int16_t)((t * KO_Q + (1u << (d - 1))) >> d);
  }
}

void ko_poly_frommsg(int16_t r[256], const uint8_t msg[32])
{
  for (int i = 0; i < 256; i++) r[i] = ((msg[i >> 3] >> (i & 7)) & 1) ? (KO_Q + 1) / 2 : 0;
}

void ko_poly_tomsg(uint8_t msg[32], const int16_t a[256])
{
  memset(msg, 0, 32);
  for (int i = 0; i < 256; i++) {
    uint32_t t = ((((uint32_t)canon(a[i]) << 1) + KO_Q / 2) / KO_Q) & 1;
    msg[i >> 3] |= (uint8_t)(t << (i & 7));
  }
}

/* ------------------------------------------------------------------ */
/* IND-CPA PKE, ref/indcpa.c:205-329                                    */
/* ------------------------------------------------------------------ */
#define KMAX 4

static void cpa_keypair_seeded(const ko_params *p, uint8_t *pk, uint8_t *sk, const uint8_t *gin, size_t ginlen)
{
  const int k = p->k;
  uint8_t g[64];
  int16_t a[KMAX * KMAX * 256], s[KMAX * 256], e[KMAX * 256], t[256];
  hash_g(p, g, gin, ginlen);                /* rho = g[0..32), sigma = g[32..64) */
  ko_gen_matrix(p, a, g, 0, NULL);
  for (int i = 0; i < k; i++) ko_getnoise(p, p->eta1, s + 256 * i, g + 32, (uint8_t)i);
  for (int i = 0; i < k; i++) ko_getnoise(p, p->eta1, e + 256 * i, g + 32, (uint8_t)(k + i));
  for (int i = 0; i < k; i++) { ko_poly_ntt(s + 256 * i); ko_poly_ntt(e + 256 * i); }
  for (int i = 0; i < k; i++) {
    ko_polyvec_basemul_acc(k, t, a + 256 * k * i, s);
    ko_poly_tomont(t);
    for (int j = 0; j < 256; j++) t[j] = ko_barrett_reduce((int16_t)(t[j] + e[256 * i + j]));
    ko_poly_tobytes(pk + 384 * i, t);
    ko_poly_tobytes(sk + 384 * i, s + 256 * i);
  }
  memcpy(pk + 384 * k, g, 32);
}

static void cpa_keypair(const ko_params *p, uint8_t *pk, uint8_t *sk, const uint8_t d[32]) { cpa_keypair_seeded(p, pk, sk, d, 32); }

static void cpa_enc(const ko_params *p, uint8_t *ct, const uint8_t m[32], const uint8_t *pk, const uint8_t coins[32])
{
  const int k = p->k;
  int16_t at[KMAX * KMAX * 256], tp[KMAX * 256], sp[KMAX * 256], w[256], n[256], mp[256];
  const uint8_t *rho = pk + 384 * k;
  for (int i = 0; i < k; i++) ko_poly_frombytes(tp + 256 * i, pk + 384 * i);
  ko_poly_frommsg(mp, m);
  ko_gen_matrix(p, at, rho, 1, NULL);
  for (int i = 0; i < k; i++) { ko_getnoise(p, p->eta1, sp + 256 * i, coins, (uint8_t)i); ko_poly_ntt(sp + 256 * i); }
  for (int i = 0; i < k; i++) {             /* u_i = invntt(At_i . sp) + e1_i */
    ko_polyvec_basemul_acc(k, w, at + 256 * k * i, sp);
    ko_invntt_tomont(w);
    ko_getnoise(p, 2, n, coins, (uint8_t)(k + i));
    for (int j = 0; j < 256; j++) w[j] = ko_barrett_reduce((int16_t)(w[j] + n[j]));
    ko_poly_compress(p->du, ct + 32 * p->du * i, w);
  }
  ko_polyvec_basemul_acc(k, w, tp, sp);     /* v = invntt(t . sp) + e2 + m */
  ko_invntt_tomont(w);
  ko_getnoise(p, 2, n, coins, (uint8_t)(2 * k));
  for (int j = 0; j < 256; j++) w[j] = ko_barrett_reduce((int16_t)(w[j] + n[j] + mp[j]));
  ko_poly_compress(p->dv, ct + ucomp_bytes(p), w);
}

static void cpa_dec(const ko_params *p, uint8_t m[32], const uint8_t *ct, const uint8_t *sk)
{
  const int k = p->k;
  int16_t u[KMAX * 256], s[KMAX * 256], v[256], w[256];
  for (int i = 0; i < k; i++) {
    ko_poly_decompress(p->du, u + 256 * i, ct + 32 * p->du * i);
    ko_poly_ntt(u + 256 * i);
    ko_poly_frombytes(s + 256 * i, sk + 384 * i);
  }
  ko_poly_decompress(p->dv, v, ct + ucomp_bytes(p));
  ko_polyvec_basemul_acc(k, w, s, u);
  ko_invntt_tomont(w);
  for (int j = 0; j < 256; j++) w[j] = ko_barrett_reduce((int16_t)(v[j] - w[j]));
  ko_poly_tomsg(m, w);
}

/* ------------------------------------------------------------------ */
/* CCA KEM, ref/kem.c:23-127                                            */
/* ------------------------------------------------------------------ */
void ko_kem_keypair(const ko_params *p, uint8_t *pk, uint8_t *sk, const uint8_t coins[64])
{
  const size_t pkb = ko_pk_bytes(p), cpa = polyvec_bytes(p);
  cpa_keypair(p, pk, sk, coins);
  memcpy(sk + cpa, pk, pkb);
  hash_h(p, sk + cpa + pkb, pk, pkb);
  memcpy(sk + cpa + pkb + 32, coins + 32, 32);
}

void ko_kem_enc(const ko_params *p, uint8_t *ct, uint8_t *ss, const uint8_t *pk, const uint8_t coins[32])
{
  uint8_t buf[64], kr[64];
  hash_h(p, buf, coins, 32);                       /* m = H(m0)          kem.c:61 */
  hash_h(p, buf + 32, pk, ko_pk_bytes(p));         /* H(pk)              kem.c:64 */
  hash_g(p, kr, buf, 64);                          /* (Kbar, r)          kem.c:65 */
  cpa_enc(p, ct, buf, pk, kr + 32);
  hash_h(p, kr + 32, ct, ko_ct_bytes(p));          /* H(ct)              kem.c:71 */
  kdf(p, ss, kr);                                  /*                    kem.c:73 */
}

void ko_kem_dec(const ko_params *p, uint8_t *ss, const uint8_t *ct, const uint8_t *sk)
{
  const size_t pkb = ko_pk_bytes(p), ctb = ko_ct_bytes(p), cpa = polyvec_bytes(p);
  const uint8_t *pk = sk + cpa, *hpk = pk + pkb, *z = hpk + 32;
  uint8_t buf[64], kr[64], cmp[4 * 352 + 160];
  uint8_t diff = 0;
  cpa_dec(p, buf, ct, sk);
  memcpy(buf + 32, hpk, 32);
  hash_g(p, kr, buf, 64);
  cpa_enc(p, cmp, buf, pk, kr + 32);
  for (size_t i = 0; i < ctb; i++) diff |= ct[i] ^ cmp[i];        /* verify, ref/verify.c:16-25 */
  hash_h(p, kr + 32, ct, ctb);
  uint8_t mask = (uint8_t)-(uint8_t)(diff != 0);                  /* cmov,   ref/verify.c:40-47 */
  for (int i = 0; i < 32; i++) kr[i] ^= mask & (kr[i] ^ z[i]);
  kdf(p, ss, kr);
}

/* ------------------------------------------------------------------ */
/* ML-KEM (FIPS 203) -- NOT in the reference, PARITY UNPINNED.          */
/* Restated from the published standard: algorithms 13-15 (K-PKE) are   */
/* the functions above (ref/indcpa.c:205-329) except that KeyGen hashes */
/* d || k; algorithms 16-18 are the three functions below.  Checked     */
/* only against tests/mlkem_model.py, an independent reading of the     */
/* same text; no official vectors are available offline.                */
/* ------------------------------------------------------------------ */
void ko_mlkem_keypair(const ko_params *p, uint8_t *ek, uint8_t *dk, const uint8_t coins[64])
{ /* ML-KEM.KeyGen_internal(d, z), FIPS 203 algorithm 16 */
  const size_t pkb = ko_pk_bytes(p), cpa = polyvec_bytes(p);
  uint8_t gin[33];
  memcpy(gin, coins, 32);
  gin[32] = (uint8_t)p->k;                                  /* algorithm 13 line 1: G(d || k) */
  cpa_keypair_seeded(p, ek, dk, gin, 33);
  memcpy(dk + cpa, ek, pkb);
  ko_sha3_256(dk + cpa + pkb, ek, pkb);
  memcpy(dk + cpa + pkb + 32, coins + 32, 32);
}

void ko_mlkem_enc(const ko_params *p, uint8_t *ct, uint8_t *ss, const uint8_t *ek, const uint8_t m[32])
{ /* ML-KEM.Encaps_internal(ek, m), algorithm 17 */
  uint8_t buf[64], kr[64];
  memcpy(buf, m, 32);
  ko_sha3_256(buf + 32, ek, ko_pk_bytes(p));
  ko_sha3_512(kr, buf, 64);                                 /* (K, r) = G(m || H(ek)) */
  cpa_enc(p, ct, m, ek, kr + 32);
  memcpy(ss, kr, 32);
}

void ko_mlkem_dec(const ko_params *p, uint8_t *ss, const uint8_t *ct, const uint8_t *dk)
{ /* ML-KEM.Decaps_internal(dk, c), algorithm 18 */
  const size_t pkb = ko_pk_bytes(p), ctb = ko_ct_bytes(p), cpa = polyvec_bytes(p);
  const uint8_t *ek = dk + cpa, *h = ek + pkb, *z = h + 32;
  uint8_t buf[64], kr[64], cmp[4 * 352 + 160], jin[32 + 4 * 352 + 160], kbar[32];
  uint8_t diff = 0;
  cpa_dec(p, buf, ct, dk);
  memcpy(buf + 32, h, 32);
  ko_sha3_512(kr, buf, 64);                                 /* (K', r') = G(m' || h) */
  memcpy(jin, z, 32);
  memcpy(jin + 32, ct, ctb);
  ko_shake256(kbar, 32, jin, 32 + ctb);                     /* Kbar = J(z || c) */
  cpa_enc(p, cmp, buf, ek, kr + 32);
  for (size_t i = 0; i < ctb; i++) diff |= ct[i] ^ cmp[i];
  uint8_t mask = (uint8_t)-(uint8_t)(diff != 0);
  for (int i = 0; i < 32; i++) ss[i] = kr[i] ^ (mask & (kr[i] ^ kbar[i]));
}

void ko_mlkem_batch_keypair(const ko_params *p, size_t n, uint8_t *ek, uint8_t *dk, const uint8_t *coins)
{ for (size_t i = 0; i < n; i++) ko_mlkem_keypair(p, ek + i * ko_pk_bytes(p), dk + i * ko_sk_bytes(p), coins + 64 * i); }
void ko_mlkem_batch_enc(const ko_params *p, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *ek, const uint8_t *coins)
{ for (size_t i = 0; i < n; i++) ko_mlkem_enc(p, ct + i * ko_ct_bytes(p), ss + 32 * i, ek + i * ko_pk_bytes(p), coins + 32 * i); }
void ko_mlkem_batch_dec(const ko_params *p, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *dk)
{ for (size_t i = 0; i < n; i++) ko_mlkem_dec(p, ss + 32 * i, ct + i * ko_ct_bytes(p), dk + i * ko_sk_bytes(p)); }

/* ------------------------------------------------------------------ */
/* key exchange, ref/kex.c:6-54 (coins replace the randombytes draws of   */
/* the KEM calls inside, in call order)                                   */
/* ------------------------------------------------------------------ */
void ko_kex_uake_initA(const ko_params *p, uint8_t *send, uint8_t *tk, uint8_t *sk, const uint8_t *pkb,
                       const uint8_t coins[96])
{
  ko_kem_keypair(p, send, sk, coins);                      /* kex.c:8 */
  ko_kem_enc(p, send + ko_pk_bytes(p), tk, pkb, coins + 64); /* kex.c:9 */
}

void ko_kex_uake_sharedB(const ko_params *p, uint8_t *send, uint8_t *k, const uint8_t *recv, const uint8_t *skb,
                         const uint8_t coins[32])
{
  uint8_t buf[64];
  ko_kem_enc(p, send, buf, recv, coins);                   /* kex.c:15 */
  ko_kem_dec(p, buf + 32, recv + ko_pk_bytes(p), skb);     /* kex.c:16 */
  kdf_n(p, k, buf, 64);
}

void ko_kex_uake_sharedA(const ko_params *p, uint8_t *k, const uint8_t *recv, const uint8_t *tk, const uint8_t *sk)
{
  uint8_t buf[64];
  ko_kem_dec(p, buf, recv, sk);                            /* kex.c:24 */
  memcpy(buf + 32, tk, 32);
  kdf_n(p, k, buf, 64);
}

void ko_kex_ake_initA(const ko_params *p, uint8_t *send, uint8_t *tk, uint8_t *sk, const uint8_t *pkb,
                      const uint8_t coins[96])
{
  ko_kex_uake_initA(p, send, tk, sk, pkb, coins);          /* kex.c:30-34, same body */
}

void ko_kex_ake_sharedB(const ko_params *p, uint8_t *send, uint8_t *k, const uint8_t *recv, const uint8_t *skb,
                        const uint8_t *pka, const uint8_t coins[64])
{
  uint8_t buf[96];
  ko_kem_enc(p, send, buf, recv, coins);                             /* kex.c:39 */
  ko_kem_enc(p, send + ko_ct_bytes(p), buf + 32, pka, coins + 32);   /* kex.c:40 */
  ko_kem_dec(p, buf + 64, recv + ko_pk_bytes(p), skb);               /* kex.c:41 */
  kdf_n(p, k, buf, 96);
}

void ko_kex_ake_sharedA(const ko_params *p, uint8_t *k, const uint8_t *recv, const uint8_t *tk, const uint8_t *sk,
                        const uint8_t *ska)
{
  uint8_t buf[96];
  ko_kem_dec(p, buf, recv, sk);                            /* kex.c:49 */
  ko_kem_dec(p, buf + 32, recv + ko_ct_bytes(p), ska);     /* kex.c:50 */
  memcpy(buf + 64, tk, 32);
  kdf_n(p, k, buf, 96);
}

void ko_batch_keypair(const ko_params *p, size_t n, uint8_t *pk, uint8_t *sk, const uint8_t *coins)
{
  const size_t pkb = ko_pk_bytes(p), skb = ko_sk_bytes(p);
  for (size_t i = 0; i < n; i++) ko_kem_keypair(p, pk + i * pkb, sk + i * skb, coins + i * 64);
}
void ko_batch_enc(const ko_params *p, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *pk, const uint8_t *coins)
{
  const size_t pkb = ko_pk_bytes(p), ctb = ko_ct_bytes(p);
  for (size_t i = 0; i < n; i++) ko_kem_enc(p, ct + i * ctb, ss + i * 32, pk + i * pkb, coins + i * 32);
}
void ko_batch_dec(const ko_params *p, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *sk)
{
  const size_t skb = ko_sk_bytes(p), ctb = ko_ct_bytes(p);
  for (size_t i = 0; i < n; i++) ko_kem_dec(p, ss + i * 32, ct + i * ctb, sk + i * skb);
}

/* ------------------------------------------------------------------ */
/* surf, ref/test_vectors.c:12-55                                       */
/* ------------------------------------------------------------------ */
void ko_surf_init(ko_surf *s) { memset(s, 0, sizeof *s); }

static void surf_round(ko_surf *s)
{
  /* the harness's 32-word seed is the first 32 decimal digits of pi */
  static const uint8_t pi[32] = {3,1,4,1,5,9,2,6,5,3,5,8,9,7,9,3,2,3,8,4,6,2,6,4,3,3,8,3,2,7,9,5};
  static const unsigned rot[4] = {5, 7, 9, 13};
  uint32_t t[12], x, sum = 0;
  for (int i = 0; i < 12; i++) t[i] = s->in[i] ^ pi[12 + i];
  for (int i = 0; i < 8; i++) s->out[i] = pi[24 + i];
  x = t[11];
  for (int loop = 0; loop < 2; loop++) {
    for (int r = 0; r < 16; r++) {
      sum += 0x9e3779b9u;
      for (int i = 0; i < 12; i++) {
        unsigned b = rot[i & 3];
        t[i] += ((x ^ pi[i]) + sum) ^ ((x << b) | (x >> (32 - b)));
        x = t[i];
      }
    }
    for (int i = 0; i < 8; i++) s->out[i] ^= t[i + 4];
  }
}

void ko_surf_bytes(ko_surf *s, uint8_t *x, size_t xlen)
{
  while (xlen--) {
    if (!s->outleft) {
      for (int i = 0; i < 4; i++) if (++s->in[i]) break;
      surf_round(s);
      s->outleft = 8;
    }
    *x++ = (uint8_t)s->out[--s->outleft];
  }
}
