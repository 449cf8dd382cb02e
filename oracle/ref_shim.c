/*
 * ref_shim.c -- glue that turns ONE parameter set of the reference (its ref/
 * or avx2/ sources, compiled where they lie under /root/reference by
 * oracle/Makefile) into a shared object with explicit-coin batch entry points.
 * TEST INFRASTRUCTURE ONLY; output goes to oracle/_ref/ and is never committed.
 *
 * The reference draws entropy through a link-time `randombytes` symbol
 * (ref/randombytes.h:7); the harnesses swap it by linking their own
 * (ref/test_vectors.c:41).  This shim does the same: its randombytes hands out
 * the caller's coins in the order the KEM asks for them (d, z | m0).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "kem.h"          /* the reference's header, found through -I */
#include "randombytes.h"
#include "kex.h"

static __thread const uint8_t *coin_cursor;

void randombytes(uint8_t *out, size_t outlen)
{
  memcpy(out, coin_cursor, outlen);
  coin_cursor += outlen;
}

size_t refshim_pk_bytes(void) { return CRYPTO_PUBLICKEYBYTES; }
size_t refshim_sk_bytes(void) { return CRYPTO_SECRETKEYBYTES; }
size_t refshim_ct_bytes(void) { return CRYPTO_CIPHERTEXTBYTES; }
const char *refshim_name(void) { return CRYPTO_ALGNAME; }

int refshim_batch_keypair(size_t n, uint8_t *pk, uint8_t *sk, const uint8_t *coins)
{
  for (size_t i = 0; i < n; i++) {
    coin_cursor = coins + 64 * i;
    crypto_kem_keypair(pk + i * CRYPTO_PUBLICKEYBYTES, sk + i * CRYPTO_SECRETKEYBYTES);
  }
  return 0;
}

int refshim_batch_enc(size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *pk, const uint8_t *coins)
{
  for (size_t i = 0; i < n; i++) {
    coin_cursor = coins + 32 * i;
    crypto_kem_enc(ct + i * CRYPTO_CIPHERTEXTBYTES, ss + i * CRYPTO_BYTES, pk + i * CRYPTO_PUBLICKEYBYTES);
  }
  return 0;
}

int refshim_batch_dec(size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *sk)
{
  for (size_t i = 0; i < n; i++)
    crypto_kem_dec(ss + i * CRYPTO_BYTES, ct + i * CRYPTO_CIPHERTEXTBYTES, sk + i * CRYPTO_SECRETKEYBYTES);
  return 0;
}

/* the reference's key exchange (ref/kex.c), one instance, coins in call order */
void refshim_kex_uake_initA(uint8_t *send, uint8_t *tk, uint8_t *sk, const uint8_t *pkb, const uint8_t *coins)
{ coin_cursor = coins; kex_uake_initA(send, tk, sk, pkb); }
void refshim_kex_uake_sharedB(uint8_t *send, uint8_t *k, const uint8_t *recv, const uint8_t *skb, const uint8_t *coins)
{ coin_cursor = coins; kex_uake_sharedB(send, k, recv, skb); }
void refshim_kex_uake_sharedA(uint8_t *k, const uint8_t *recv, const uint8_t *tk, const uint8_t *sk)
{ kex_uake_sharedA(k, recv, tk, sk); }
void refshim_kex_ake_initA(uint8_t *send, uint8_t *tk, uint8_t *sk, const uint8_t *pkb, const uint8_t *coins)
{ coin_cursor = coins; kex_ake_initA(send, tk, sk, pkb); }
void refshim_kex_ake_sharedB(uint8_t *send, uint8_t *k, const uint8_t *recv, const uint8_t *skb, const uint8_t *pka, const uint8_t *coins)
{ coin_cursor = coins; kex_ake_sharedB(send, k, recv, skb, pka); }
void refshim_kex_ake_sharedA(uint8_t *k, const uint8_t *recv, const uint8_t *tk, const uint8_t *sk, const uint8_t *ska)
{ kex_ake_sharedA(k, recv, tk, sk, ska); }
