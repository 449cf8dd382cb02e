/*
 * test_vectors_hip -- the reference's golden-vector stream produced by the
 * GPU library.
 *
 *   test_vectors_hip <512|768|1024>[-90s] [iterations=10000] | sha256sum
 *   test_vectors_hip <set> <iterations> --coins     (PRNG self-check: prints d, z, m0 only, no GPU)
 *
 * prints byte for byte what ref/test_vectors.c:57-105 prints for the same
 * parameter set, so the digest can be checked against the reference's
 * SHA256SUMS (harness/check.sh does that for all six sets).  The reference
 * draws d, z and m0 from its surf PRNG one KEM call at a time; none of the KEM
 * outputs feed back into the PRNG, so all coins are drawn first and the
 * library runs three batched calls.
 */
#include <stdlib.h>
#include "common.h"
#include "kyber_batch.h"

int main(int argc, char **argv)
{
  int k, n90, coins_only = 0;
  long n, i;
  if (argc >= 3 && !strcmp(argv[argc - 1], "--coins")) { coins_only = 1; argc--; }
  if (parse_set(argc, argv, &k, &n90, &n, 10000) || n <= 0) {
    fprintf(stderr, "usage: %s <512|768|1024>[-90s] [iterations]\n", argv[0]);
    return 2;
  }
  const size_t pkb = kyber_hip_pk_bytes(k), skb = kyber_hip_sk_bytes(k), ctb = kyber_hip_ct_bytes(k);
  uint8_t *kc = malloc((size_t)n * 64), *ec = malloc((size_t)n * 32);
  uint8_t *pk = malloc((size_t)n * pkb), *sk = malloc((size_t)n * skb), *ct = malloc((size_t)n * ctb);
  uint8_t *ssb = malloc((size_t)n * 32), *ssa = malloc((size_t)n * 32);
  if (!kc || !ec || !pk || !sk || !ct || !ssb || !ssa) return 3;

  surf_t rng;
  surf_init(&rng);
  for (i = 0; i < n; i++) {
    surf_bytes(&rng, kc + 64 * i, 64); /* d (ref/indcpa.c:215) then z (ref/kem.c:32) */
    surf_bytes(&rng, ec + 32 * i, 32); /* m0 (ref/kem.c:59) */
  }
  if (coins_only) {
    for (i = 0; i < n; i++) {
      put_hex(stdout, "", kc + 64 * i, 64, 0);
      put_hex(stdout, "", ec + 32 * i, 32, 0);
    }
    return 0;
  }
  int rc;
  if ((rc = kyber_batch_keypair(k, n90, (size_t)n, pk, sk, kc)) || (rc = kyber_batch_enc(k, n90, (size_t)n, ct, ssb, pk, ec)) ||
      (rc = kyber_batch_dec(k, n90, (size_t)n, ssa, ct, sk))) {
    fprintf(stderr, "kyber_batch_* failed: %d (%s)\n", rc, kyber_hip_last_error());
    return 1;
  }
  for (i = 0; i < n; i++) {
    put_hex(stdout, "", kc + 64 * i, 32, 0);
    put_hex(stdout, "", kc + 64 * i + 32, 32, 0);
    put_hex(stdout, "Public Key: ", pk + pkb * i, pkb, 0);
    put_hex(stdout, "Secret Key: ", sk + skb * i, skb, 0);
    put_hex(stdout, "", ec + 32 * i, 32, 0);
    put_hex(stdout, "Ciphertext: ", ct + ctb * i, ctb, 0);
    put_hex(stdout, "Shared Secret B: ", ssb + 32 * i, 32, 0);
    put_hex(stdout, "Shared Secret A: ", ssa + 32 * i, 32, 0);
    if (memcmp(ssa + 32 * i, ssb + 32 * i, 32)) {
      fprintf(stderr, "ERROR\n"); /* ref/test_vectors.c:99-102 */
      return 1;
    }
  }
  return 0;
}
