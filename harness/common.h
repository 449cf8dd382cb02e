/*
 * harness/common.h -- deterministic byte sources and the hex writer shared by
 * the two vector harnesses.  Stand-alone C: nothing here touches oracle/.
 *
 *   surf_*  : the counter-mode PRNG ref/test_vectors.c:12-55 substitutes for
 *             randombytes (D. J. Bernstein's SURF on the digits of pi).
 *   drbg_*  : the AES-256 CTR_DRBG of ref/rng.c:139-213 (NIST KAT generator),
 *             with a small table-free AES-256 block encryption.
 */
#ifndef KYBER_HARNESS_COMMON_H
#define KYBER_HARNESS_COMMON_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- surf */
typedef struct {
  uint32_t ctr[12]; /* 384-bit little-endian block counter (only 4 words ever move) */
  uint32_t blk[8];  /* current output block, consumed from the top */
  int left;
} surf_t;

static void surf_block(surf_t *s)
{
  static const uint8_t seed[32] = {3,1,4,1,5,9,2,6,5,3,5,8,9,7,9,3,2,3,8,4,6,2,6,4,3,3,8,3,2,7,9,5};
  uint32_t t[12], x, sum = 0;
  int i, r, pass;
  for (i = 0; i < 12; i++) t[i] = s->ctr[i] ^ seed[12 + i];
  for (i = 0; i < 8; i++) s->blk[i] = seed[24 + i];
  x = t[11];
  for (pass = 0; pass < 2; pass++) {
    for (r = 0; r < 16; r++) {
      sum += 0x9e3779b9u;
      for (i = 0; i < 12; i++) {
        static const int sh[4] = {5, 7, 9, 13};
        const int b = sh[i & 3];
        x = t[i] += ((x ^ seed[i]) + sum) ^ (x << b | x >> (32 - b));
      }
    }
    for (i = 0; i < 8; i++) s->blk[i] ^= t[i + 4];
  }
}

static void surf_init(surf_t *s) { memset(s, 0, sizeof *s); }

static void surf_bytes(surf_t *s, uint8_t *out, size_t n)
{
  while (n--) {
    if (!s->left) {
      int i;
      for (i = 0; i < 4 && ++s->ctr[i] == 0; i++) {}
      surf_block(s);
      s->left = 8;
    }
    *out++ = (uint8_t)s->blk[--s->left];
  }
}

/* ---------------------------------------------------------------- AES-256 (one block) */
static uint8_t aes_sbox[256];

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
  int a, i;
  if (aes_sbox[0]) return;
  for (a = 0; a < 256; a++) {
    uint8_t inv = 0, y;
    if (a) { /* a^254 = a^-1 in GF(2^8) */
      uint8_t p = (uint8_t)a;
      inv = 1;
      for (i = 0; i < 7; i++) { p = gf_mul(p, p); inv = gf_mul(inv, p); }
    }
    y = inv;
    for (i = 1; i <= 4; i++) y ^= (uint8_t)(inv << i | inv >> (8 - i));
    aes_sbox[a] = y ^ 0x63;
  }
}

static void aes256_block(const uint8_t key[32], const uint8_t in[16], uint8_t out[16])
{
  uint8_t rk[15][16], st[16], t[16];
  uint8_t rcon = 1;
  int i, r, c;
  aes_setup();
  memcpy(rk, key, 32);
  for (i = 8; i < 60; i++) { /* words of the expanded key */
    uint8_t *w = &rk[0][0] + 4 * i, *p = w - 4, *q = w - 32, v[4];
    memcpy(v, p, 4);
    if (i % 8 == 0) {
      uint8_t a0 = v[0];
      v[0] = aes_sbox[v[1]] ^ rcon; v[1] = aes_sbox[v[2]]; v[2] = aes_sbox[v[3]]; v[3] = aes_sbox[a0];
      rcon = gf_mul(rcon, 2);
    } else if (i % 8 == 4) {
      for (c = 0; c < 4; c++) v[c] = aes_sbox[v[c]];
    }
    for (c = 0; c < 4; c++) w[c] = q[c] ^ v[c];
  }
  for (i = 0; i < 16; i++) st[i] = in[i] ^ rk[0][i];
  for (r = 1; r <= 14; r++) {
    for (c = 0; c < 4; c++) /* SubBytes + ShiftRows: column c takes row j from column c+j */
      for (i = 0; i < 4; i++) t[4 * c + i] = aes_sbox[st[4 * ((c + i) & 3) + i]];
    if (r < 14) {
      for (c = 0; c < 4; c++) {
        const uint8_t *a = t + 4 * c;
        const uint8_t all = a[0] ^ a[1] ^ a[2] ^ a[3];
        for (i = 0; i < 4; i++) st[4 * c + i] = a[i] ^ all ^ gf_mul(a[i] ^ a[(i + 1) & 3], 2);
      }
    } else {
      memcpy(st, t, 16);
    }
    for (i = 0; i < 16; i++) st[i] ^= rk[r][i];
  }
  memcpy(out, st, 16);
}

/* ---------------------------------------------------------------- CTR_DRBG */
typedef struct { uint8_t key[32], v[16]; } drbg_t;

static void drbg_step(drbg_t *g, uint8_t out[16])
{
  int i;
  for (i = 15; i >= 0 && ++g->v[i] == 0; i--) {}
  aes256_block(g->key, g->v, out);
}

static void drbg_update(drbg_t *g, const uint8_t *data48)
{
  uint8_t tmp[48];
  int i;
  for (i = 0; i < 3; i++) drbg_step(g, tmp + 16 * i);
  if (data48) for (i = 0; i < 48; i++) tmp[i] ^= data48[i];
  memcpy(g->key, tmp, 32);
  memcpy(g->v, tmp + 32, 16);
}

static void drbg_init(drbg_t *g, const uint8_t entropy48[48])
{
  memset(g, 0, sizeof *g);
  drbg_update(g, entropy48);
}

static void drbg_bytes(drbg_t *g, uint8_t *out, size_t n)
{
  uint8_t blk[16];
  while (n) {
    size_t take = n < 16 ? n : 16;
    drbg_step(g, blk);
    memcpy(out, blk, take);
    out += take;
    n -= take;
  }
  drbg_update(g, NULL);
}

/* ---------------------------------------------------------------- output */
static void put_hex(FILE *f, const char *label, const uint8_t *b, size_t n, int upper)
{
  const char *dig = upper ? "0123456789ABCDEF" : "0123456789abcdef";
  static char line[2 * 3200 + 64];
  size_t o = 0, i;
  while (*label) line[o++] = *label++;
  for (i = 0; i < n; i++) { line[o++] = dig[b[i] >> 4]; line[o++] = dig[b[i] & 15]; }
  line[o++] = '\n';
  fwrite(line, 1, o, f);
}

static int parse_set(int argc, char **argv, int *k, int *nineties, long *count, long dflt)
{
  *k = 0; *nineties = 0; *count = dflt;
  if (argc >= 2) {
    if (!strncmp(argv[1], "512", 3)) *k = 2;
    else if (!strncmp(argv[1], "768", 3)) *k = 3;
    else if (!strncmp(argv[1], "1024", 4)) *k = 4;
    if (strstr(argv[1], "-90s")) *nineties = 1;
  }
  if (argc >= 3) *count = atol(argv[2]);
  return *k ? 0 : -1;
}
#endif
