/*
 * kyber_oracle.h -- CPU restatement of the round-3 Kyber KEM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (kyber_amd/, include/)
 * links, loads or calls this code.  It exists so that tests/, the smoke check
 * and bench.py's cpu_baseline leg have a checker that runs on a CPU.
 *
 * Parity pin: this restatement reproduces all six lines of the reference's
 * SHA256SUMS (10 000 keypair/enc/dec each, tests/test_oracle_vectors.py) and
 * is cross-checked against oracle/_ref (the reference's own ref/ and avx2/
 * sources compiled in place) on random coins (tests/test_oracle_vs_ref.py).
 *
 * One translation unit, parameterised at run time instead of the reference's
 * -DKYBER_K / -DKYBER_90S compile-time switches (ref/params.h:4-8).
 */
#ifndef KYBER_ORACLE_H
#define KYBER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KO_N 256
#define KO_Q 3329
#define KO_SYMBYTES 32

typedef struct {
  int k;        /* 2, 3, 4                     ref/params.h:11-31 */
  int eta1;     /* 3 for k=2 else 2            ref/params.h:42-54 */
  int du;       /* 10, 10, 11                  ref/params.h:45,49,53 */
  int dv;       /* 4, 4, 5                     ref/params.h:44,48,52 */
  int nineties; /* 0 = SHAKE/SHA3 set, 1 = AES-256-CTR/SHA-2 set  ref/symmetric.h:8-58 */
} ko_params;

/* fills *p; returns 0, or -1 for a k outside {2,3,4} */
int ko_params_init(ko_params *p, int k, int nineties);

size_t ko_pk_bytes(const ko_params *p); /* ref/params.h:63 */
size_t ko_sk_bytes(const ko_params *p); /* ref/params.h:65 */
size_t ko_ct_bytes(const ko_params *p); /* ref/params.h:66 */

/* KEM (ref/kem.c:23-127) with the three randombytes() draws made explicit:
 * keypair coins = d(32) || z(32)   (ref/indcpa.c:215, ref/kem.c:32)
 * enc coins     = m0(32)           (ref/kem.c:59)                           */
void ko_kem_keypair(const ko_params *p, uint8_t *pk, uint8_t *sk, const uint8_t coins[64]);
void ko_kem_enc(const ko_params *p, uint8_t *ct, uint8_t *ss, const uint8_t *pk, const uint8_t coins[32]);
void ko_kem_dec(const ko_params *p, uint8_t *ss, const uint8_t *ct, const uint8_t *sk);

/* batched helpers (plain loops) used as the CPU side of parity tests and as
 * the "port" CPU baseline; instance i lives at base + i*<bytes>. */
/* key exchange (ref/kex.c:6-54, ref/kex.h:7-29); one instance per call.
 * coins: initA d||z||m0 (96), uake_sharedB m0 (32), ake_sharedB m0||m0' (64). */
void ko_kex_uake_initA(const ko_params *p, uint8_t *send, uint8_t *tk, uint8_t *sk, const uint8_t *pkb, const uint8_t coins[96]);
void ko_kex_uake_sharedB(const ko_params *p, uint8_t *send, uint8_t *k, const uint8_t *recv, const uint8_t *skb, const uint8_t coins[32]);
void ko_kex_uake_sharedA(const ko_params *p, uint8_t *k, const uint8_t *recv, const uint8_t *tk, const uint8_t *sk);
void ko_kex_ake_initA(const ko_params *p, uint8_t *send, uint8_t *tk, uint8_t *sk, const uint8_t *pkb, const uint8_t coins[96]);
void ko_kex_ake_sharedB(const ko_params *p, uint8_t *send, uint8_t *k, const uint8_t *recv, const uint8_t *skb, const uint8_t *pka,
                        const uint8_t coins[64]);
void ko_kex_ake_sharedA(const ko_params *p, uint8_t *k, const uint8_t *recv, const uint8_t *tk, const uint8_t *sk, const uint8_t *ska);

/* ML-KEM (FIPS 203) internal algorithms 16-18 -- not in the reference, parity unpinned (see kyber_oracle.c) */
void ko_mlkem_keypair(const ko_params *p, uint8_t *ek, uint8_t *dk, const uint8_t coins[64]);
void ko_mlkem_enc(const ko_params *p, uint8_t *ct, uint8_t *ss, const uint8_t *ek, const uint8_t m[32]);
void ko_mlkem_dec(const ko_params *p, uint8_t *ss, const uint8_t *ct, const uint8_t *dk);
void ko_mlkem_batch_keypair(const ko_params *p, size_t n, uint8_t *ek, uint8_t *dk, const uint8_t *coins);
void ko_mlkem_batch_enc(const ko_params *p, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *ek, const uint8_t *coins);
void ko_mlkem_batch_dec(const ko_params *p, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *dk);

void ko_batch_keypair(const ko_params *p, size_t n, uint8_t *pk, uint8_t *sk, const uint8_t *coins);
void ko_batch_enc(const ko_params *p, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *pk, const uint8_t *coins);
void ko_batch_dec(const ko_params *p, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *sk);

/* ---- stage-level entry points (for per-kernel parity tests) ---- */
int16_t ko_montgomery_reduce(int32_t a);          /* ref/reduce.c:16-23 */
int16_t ko_barrett_reduce(int16_t a);             /* ref/reduce.c:35-42 */
const int16_t *ko_zetas(void);                    /* ref/ntt.c:39-56 (computed, ref/ntt.c:21-36) */
void ko_ntt(int16_t r[256]);                      /* ref/ntt.c:80-95  (no final reduce)  */
void ko_poly_ntt(int16_t r[256]);                 /* ref/poly.c:244-248 (ntt + barrett)  */
void ko_invntt_tomont(int16_t r[256]);            /* ref/ntt.c:106-126 */
void ko_poly_basemul(int16_t r[256], const int16_t a[256], const int16_t b[256]); /* ref/poly.c:273-280 */
void ko_polyvec_basemul_acc(int k, int16_t r[256], const int16_t *a, const int16_t *b); /* ref/polyvec.c:189-201 */
void ko_poly_tomont(int16_t r[256]);              /* ref/poly.c:290-296 */
void ko_cbd(int eta, int16_t r[256], const uint8_t *buf); /* ref/cbd.c:58-107 */
void ko_poly_tobytes(uint8_t r[384], const int16_t a[256]);   /* ref/poly.c:111-126 */
void ko_poly_frombytes(int16_t r[256], const uint8_t a[384]); /* ref/poly.c:138-145 */
void ko_poly_compress(int d, uint8_t *r, const int16_t a[256]);   /* d=4,5: ref/poly.c:18-58; d=10,11: ref/polyvec.c:15-64 (per poly) */
void ko_poly_decompress(int d, int16_t r[256], const uint8_t *a); /* ref/poly.c:70-100; ref/polyvec.c:76-115 */
void ko_poly_frommsg(int16_t r[256], const uint8_t msg[32]);  /* ref/poly.c:155-170 */
void ko_poly_tomsg(uint8_t msg[32], const int16_t a[256]);    /* ref/poly.c:180-194 */
/* a[i*k+j] <- entry (i,j) of A (or A^T); ref/indcpa.c:164-192.  nblocks_out
 * (may be NULL) receives, per entry, how many XOF blocks were consumed. */
void ko_gen_matrix(const ko_params *p, int16_t *a, const uint8_t seed[32], int transposed, int *nblocks_out);
void ko_getnoise(const ko_params *p, int eta, int16_t r[256], const uint8_t seed[32], uint8_t nonce); /* ref/poly.c:208-232 */

/* ---- symmetric primitives ---- */
void ko_keccak_f1600(uint64_t s[25]);                                         /* ref/fips202.c:82-344 */
void ko_shake128(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen); /* ref/fips202.c:700-711 */
void ko_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen); /* ref/fips202.c:723-734 */
void ko_sha3_256(uint8_t h[32], const uint8_t *in, size_t inlen);             /* ref/fips202.c:745-754 */
void ko_sha3_512(uint8_t h[64], const uint8_t *in, size_t inlen);             /* ref/fips202.c:765-774 */
void ko_sha256(uint8_t h[32], const uint8_t *in, size_t inlen);               /* ref/sha256.c */
void ko_sha512(uint8_t h[64], const uint8_t *in, size_t inlen);               /* ref/sha512.c */
/* AES-256-CTR keystream: 12-byte nonce || 32-bit big-endian block counter
 * starting at `ctr0` (ref/aes256ctr.c:530-555, :565-586). */
void ko_aes256ctr(uint8_t *out, size_t outlen, const uint8_t key[32], const uint8_t nonce[12], uint32_t ctr0);

/* DJB "surf" generator exactly as the reference's test-vector harness seeds
 * and consumes it (ref/test_vectors.c:12-55).  State is caller-owned. */
typedef struct { uint32_t in[12]; uint32_t out[8]; int outleft; } ko_surf;
void ko_surf_init(ko_surf *s);
void ko_surf_bytes(ko_surf *s, uint8_t *x, size_t xlen);

#ifdef __cplusplus
}
#endif
#endif
