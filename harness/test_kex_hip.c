/*
 * test_kex_hip -- the reference's key-exchange test (ref/test_kex.c:8-67) on the
 * GPU library, n exchanges at once: static keys for Bob and Alice, then UAKE
 * and AKE; both sides must derive the same, non-zero key.  Prints the
 * reference's four size lines on success.
 *
 *   test_kex_hip <512|768|1024>[-90s] [n=256]
 */
#include <stdlib.h>
#include "common.h"
#include "kyber_batch.h"

int main(int argc, char **argv)
{
  int k, n90;
  long n, i;
  if (parse_set(argc, argv, &k, &n90, &n, 256) || n <= 0) {
    fprintf(stderr, "usage: %s <512|768|1024>[-90s] [n]\n", argv[0]);
    return 2;
  }
  const size_t pkb = kyber_hip_pk_bytes(k), skb = kyber_hip_sk_bytes(k), ctb = kyber_hip_ct_bytes(k);
  uint8_t *pk_b = malloc(n * pkb), *sk_b = malloc(n * skb), *pk_a = malloc(n * pkb), *sk_a = malloc(n * skb), *eska = malloc(n * skb);
  uint8_t *senda = malloc(n * (pkb + ctb)), *sendb = malloc(n * 2 * ctb), *tk = malloc(n * 32), *ka = malloc(n * 32), *kb = malloc(n * 32);
  uint8_t zero[32] = {0};
  if (!pk_b || !sk_b || !pk_a || !sk_a || !eska || !senda || !sendb || !tk || !ka || !kb) return 3;
  int rc = 0;
  rc |= kyber_batch_keypair(k, n90, n, pk_b, sk_b, NULL); /* static key of Bob   */
  rc |= kyber_batch_keypair(k, n90, n, pk_a, sk_a, NULL); /* static key of Alice */
  /* unilaterally authenticated key exchange */
  rc |= kyber_batch_kex_uake_initA(k, n90, n, senda, tk, eska, pk_b, NULL, 0);
  rc |= kyber_batch_kex_uake_sharedB(k, n90, n, sendb, kb, senda, sk_b, NULL, 0);
  rc |= kyber_batch_kex_uake_sharedA(k, n90, n, ka, sendb, tk, eska);
  if (rc) { fprintf(stderr, "kyber_batch_kex_* failed (%s)\n", kyber_hip_last_error()); return 1; }
  for (i = 0; i < n; i++) {
    if (memcmp(ka + 32 * i, kb + 32 * i, 32)) { printf("Error in UAKE\n"); return 1; }
    if (!memcmp(ka + 32 * i, zero, 32)) { printf("Error: UAKE produces zero key\n"); return 1; }
  }
  /* mutually authenticated key exchange */
  rc |= kyber_batch_kex_ake_initA(k, n90, n, senda, tk, eska, pk_b, NULL, 0);
  rc |= kyber_batch_kex_ake_sharedB(k, n90, n, sendb, kb, senda, sk_b, pk_a, NULL, 0);
  rc |= kyber_batch_kex_ake_sharedA(k, n90, n, ka, sendb, tk, eska, sk_a);
  if (rc) { fprintf(stderr, "kyber_batch_kex_* failed (%s)\n", kyber_hip_last_error()); return 1; }
  for (i = 0; i < n; i++) {
    if (memcmp(ka + 32 * i, kb + 32 * i, 32)) { printf("Error in AKE\n"); return 1; }
    if (!memcmp(ka + 32 * i, zero, 32)) { printf("Error: AKE produces zero key\n"); return 1; }
  }
  printf("KEX_UAKE_SENDABYTES: %d\n", (int)(pkb + ctb));
  printf("KEX_UAKE_SENDBBYTES: %d\n", (int)ctb);
  printf("KEX_AKE_SENDABYTES: %d\n", (int)(pkb + ctb));
  printf("KEX_AKE_SENDBBYTES: %d\n", (int)(2 * ctb));
  return 0;
}
