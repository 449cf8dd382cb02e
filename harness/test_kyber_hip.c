/*
 * test_kyber_hip -- the reference's functional test (ref/test_kyber.c:8-112) on
 * the GPU library: NTESTS rounds of (1) keypair/enc/dec agree, (2) a random
 * secret key gives a different shared secret, (3) a ciphertext with one byte
 * changed gives a different shared secret.  The reference runs the rounds one
 * KEM call at a time; here each step is one batched call over all rounds,
 * entropy from the library's randombytes (coins == NULL).  Additionally the
 * batch-of-one namespaced functions are exercised once per parameter set.
 *
 *   test_kyber_hip <512|768|1024>[-90s] [NTESTS=1000]
 * prints the three size lines of the reference on success, exit status 1 and
 * the reference's error line otherwise.
 */
#include <stdlib.h>
#include "common.h"
#include "kyber_api.h"
#include "kyber_batch.h"

typedef int (*kp_fn)(uint8_t *, uint8_t *);
typedef int (*enc_fn)(uint8_t *, uint8_t *, const uint8_t *);
typedef int (*dec_fn)(uint8_t *, const uint8_t *, const uint8_t *);

static int single_call_api(int k, int n90)
{
  static const kp_fn kp[6] = {pqcrystals_kyber512_hip_keypair, pqcrystals_kyber512_90s_hip_keypair, pqcrystals_kyber768_hip_keypair,
                              pqcrystals_kyber768_90s_hip_keypair, pqcrystals_kyber1024_hip_keypair, pqcrystals_kyber1024_90s_hip_keypair};
  static const enc_fn en[6] = {pqcrystals_kyber512_hip_enc, pqcrystals_kyber512_90s_hip_enc, pqcrystals_kyber768_hip_enc,
                               pqcrystals_kyber768_90s_hip_enc, pqcrystals_kyber1024_hip_enc, pqcrystals_kyber1024_90s_hip_enc};
  static const dec_fn de[6] = {pqcrystals_kyber512_hip_dec, pqcrystals_kyber512_90s_hip_dec, pqcrystals_kyber768_hip_dec,
                               pqcrystals_kyber768_90s_hip_dec, pqcrystals_kyber1024_hip_dec, pqcrystals_kyber1024_90s_hip_dec};
  const int i = 2 * (k - 2) + n90;
  static uint8_t pk[1568], sk[3168], ct[1568];
  uint8_t a[32], b[32];
  if (kp[i](pk, sk) || en[i](ct, b, pk) || de[i](a, ct, sk)) return 1;
  return memcmp(a, b, 32) != 0;
}

int main(int argc, char **argv)
{
  int k, n90;
  long n, i;
  if (parse_set(argc, argv, &k, &n90, &n, 1000) || n <= 0) {
    fprintf(stderr, "usage: %s <512|768|1024>[-90s] [NTESTS]\n", argv[0]);
    return 2;
  }
  const size_t pkb = kyber_hip_pk_bytes(k), skb = kyber_hip_sk_bytes(k), ctb = kyber_hip_ct_bytes(k);
  uint8_t *pk = malloc(n * pkb), *sk = malloc(n * skb), *ct = malloc(n * ctb), *bad = malloc(n * ctb), *rsk = malloc(n * skb);
  uint8_t *key_a = malloc(n * 32), *key_b = malloc(n * 32);
  if (!pk || !sk || !ct || !bad || !rsk || !key_a || !key_b) return 3;
  int rc;
  /* test_keys */
  if ((rc = kyber_batch_keypair(k, n90, n, pk, sk, NULL)) || (rc = kyber_batch_enc(k, n90, n, ct, key_b, pk, NULL)) ||
      (rc = kyber_batch_dec(k, n90, n, key_a, ct, sk))) {
    fprintf(stderr, "kyber_batch_* failed: %d (%s)\n", rc, kyber_hip_last_error());
    return 1;
  }
  for (i = 0; i < n; i++)
    if (memcmp(key_a + 32 * i, key_b + 32 * i, 32)) { printf("ERROR keys\n"); return 1; }
  /* test_invalid_sk_a: replace the secret key with random bytes */
  randombytes(rsk, n * skb);
  if ((rc = kyber_batch_dec(k, n90, n, key_a, ct, rsk))) return 1;
  for (i = 0; i < n; i++)
    if (!memcmp(key_a + 32 * i, key_b + 32 * i, 32)) { printf("ERROR invalid sk\n"); return 1; }
  /* test_invalid_ciphertext: change one byte at a random position */
  memcpy(bad, ct, n * ctb);
  for (i = 0; i < n; i++) {
    uint8_t b;
    size_t pos;
    do randombytes(&b, 1); while (!b);
    randombytes((uint8_t *)&pos, sizeof pos);
    bad[ctb * i + pos % ctb] ^= b;
  }
  if ((rc = kyber_batch_dec(k, n90, n, key_a, bad, sk))) return 1;
  for (i = 0; i < n; i++)
    if (!memcmp(key_a + 32 * i, key_b + 32 * i, 32)) { printf("ERROR invalid ciphertext\n"); return 1; }
  if (single_call_api(k, n90)) { printf("ERROR keys (single-call API)\n"); return 1; }
  printf("CRYPTO_SECRETKEYBYTES:  %d\n", (int)skb);
  printf("CRYPTO_PUBLICKEYBYTES:  %d\n", (int)pkb);
  printf("CRYPTO_CIPHERTEXTBYTES: %d\n", (int)ctb);
  return 0;
}
