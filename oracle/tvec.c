/*
 * tvec.c -- the reference's golden-vector stream, produced by the oracle.
 * TEST INFRASTRUCTURE ONLY.
 *
 * Emits byte for byte what ref/test_vectors.c:57-105 prints when linked with
 * one parameter set: per iteration hex(d), hex(z), "Public Key: ..",
 * "Secret Key: ..", hex(m0), "Ciphertext: ..", "Shared Secret B: ..",
 * "Shared Secret A: ..".  sha256 of 10000 iterations must equal the matching
 * line of the reference's SHA256SUMS (committed as tests/golden/SHA256SUMS).
 *
 *   tvec <k:2|3|4> <nineties:0|1> [iterations=10000]
 */
#include "kyber_oracle.h"
#include <stdio.h>
#include <stdlib.h>

static void hexline(const char *label, const uint8_t *b, size_t n)
{
  static const char dig[] = "0123456789abcdef";
  static char line[2 * 3200 + 64];
  size_t o = 0;
  while (*label) line[o++] = *label++;
  for (size_t i = 0; i < n; i++) { line[o++] = dig[b[i] >> 4]; line[o++] = dig[b[i] & 15]; }
  line[o++] = '\n';
  fwrite(line, 1, o, stdout);
}

int main(int argc, char **argv)
{
  ko_params p;
  ko_surf rng;
  static uint8_t pk[1568], sk[3168], ct[1568];
  uint8_t coins[64], m0[32], ssb[32], ssa[32];
  if (argc < 3 || ko_params_init(&p, atoi(argv[1]), atoi(argv[2]))) {
    fprintf(stderr, "usage: tvec <2|3|4> <0|1> [iterations]\n");
    return 2;
  }
  long n = argc > 3 ? atol(argv[3]) : 10000;
  ko_surf_init(&rng);
  for (long i = 0; i < n; i++) {
    ko_surf_bytes(&rng, coins, 32);      /* d,  drawn at ref/indcpa.c:215 */
    ko_surf_bytes(&rng, coins + 32, 32); /* z,  drawn at ref/kem.c:32     */
    hexline("", coins, 32);
    hexline("", coins + 32, 32);
    ko_kem_keypair(&p, pk, sk, coins);
    hexline("Public Key: ", pk, ko_pk_bytes(&p));
    hexline("Secret Key: ", sk, ko_sk_bytes(&p));
    ko_surf_bytes(&rng, m0, 32);         /* m0, drawn at ref/kem.c:59     */
    hexline("", m0, 32);
    ko_kem_enc(&p, ct, ssb, pk, m0);
    hexline("Ciphertext: ", ct, ko_ct_bytes(&p));
    hexline("Shared Secret B: ", ssb, 32);
    ko_kem_dec(&p, ssa, ct, sk);
    hexline("Shared Secret A: ", ssa, 32);
    for (int j = 0; j < 32; j++)
      if (ssa[j] != ssb[j]) { fprintf(stderr, "ERROR\n"); return 1; }
  }
  return 0;
}
