/*
 * kat_hip -- NIST known-answer responses produced by the GPU library.
 *
 *   kat_hip <512|768|1024>[-90s] [records=100] [--record0] > PQCkemKAT_<skbytes>.rsp
 *
 * writes what ref/PQCgenKAT_kem.c:50-105 writes into its .rsp file: a
 * "# <algname>" header, then per record count/seed/pk/sk/ct/ss in upper-case
 * hex.  --coins prints seed, d||z, m0 per record and exits (DRBG self-check,
 * no GPU).  --record0 prints only record 0 without the header: its sha256 is the
 * `nistkat-sha256` value of the reference's Kyber*_META.yml.  Each record's
 * 48-byte seed comes from the master CTR_DRBG seeded with 0,1,..,47 and seeds
 * the DRBG that hands d, z and m0 to the three KEM calls (ref/rng.c:139-213).
 */
#include <stdlib.h>
#include "common.h"
#include "kyber_batch.h"

int main(int argc, char **argv)
{
  int k, n90, record0 = 0, coins_only = 0;
  long n, i;
  if (argc >= 3 && !strcmp(argv[argc - 1], "--coins")) { coins_only = 1; argc--; }
  if (argc >= 3 && !strcmp(argv[argc - 1], "--record0")) { record0 = 1; argc--; }
  if (parse_set(argc, argv, &k, &n90, &n, 100) || n <= 0) {
    fprintf(stderr, "usage: %s <512|768|1024>[-90s] [records] [--record0]\n", argv[0]);
    return 2;
  }
  if (record0 && !coins_only) n = 1;
  const size_t pkb = kyber_hip_pk_bytes(k), skb = kyber_hip_sk_bytes(k), ctb = kyber_hip_ct_bytes(k);
  uint8_t *seed = malloc((size_t)n * 48), *kc = malloc((size_t)n * 64), *ec = malloc((size_t)n * 32);
  uint8_t *pk = malloc((size_t)n * pkb), *sk = malloc((size_t)n * skb), *ct = malloc((size_t)n * ctb);
  uint8_t *ss = malloc((size_t)n * 32), *ss1 = malloc((size_t)n * 32);
  if (!seed || !kc || !ec || !pk || !sk || !ct || !ss || !ss1) return 3;

  uint8_t entropy[48];
  drbg_t master, g;
  for (i = 0; i < 48; i++) entropy[i] = (uint8_t)i;
  drbg_init(&master, entropy);
  for (i = 0; i < n; i++) {
    drbg_bytes(&master, seed + 48 * i, 48);
    drbg_init(&g, seed + 48 * i);
    drbg_bytes(&g, kc + 64 * i, 32);      /* d  */
    drbg_bytes(&g, kc + 64 * i + 32, 32); /* z  */
    drbg_bytes(&g, ec + 32 * i, 32);      /* m0 */
  }
  if (coins_only) {
    for (i = 0; i < n; i++) {
      put_hex(stdout, "", seed + 48 * i, 48, 0);
      put_hex(stdout, "", kc + 64 * i, 64, 0);
      put_hex(stdout, "", ec + 32 * i, 32, 0);
    }
    return 0;
  }
  int rc;
  if ((rc = kyber_batch_keypair(k, n90, (size_t)n, pk, sk, kc)) || (rc = kyber_batch_enc(k, n90, (size_t)n, ct, ss, pk, ec)) ||
      (rc = kyber_batch_dec(k, n90, (size_t)n, ss1, ct, sk))) {
    fprintf(stderr, "kyber_batch_* failed: %d (%s)\n", rc, kyber_hip_last_error());
    return 1;
  }
  if (!record0) printf("# Kyber%d%s\n\n", 256 * k, n90 ? "-90s" : "");
  for (i = 0; i < n; i++) {
    printf("count = %ld\n", i);
    fflush(stdout);
    put_hex(stdout, "seed = ", seed + 48 * i, 48, 1);
    put_hex(stdout, "pk = ", pk + pkb * i, pkb, 1);
    put_hex(stdout, "sk = ", sk + skb * i, skb, 1);
    put_hex(stdout, "ct = ", ct + ctb * i, ctb, 1);
    put_hex(stdout, "ss = ", ss + 32 * i, 32, 1);
    if (!record0) printf("\n");
    if (memcmp(ss + 32 * i, ss1 + 32 * i, 32)) {
      fprintf(stderr, "crypto_kem_dec returned bad 'ss' value\n");
      return 1;
    }
  }
  return 0;
}
