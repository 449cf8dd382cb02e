/*
 * test_randombytes_hook -- the reference swaps its entropy source by linking
 * another definition of randombytes (ref/randombytes.h:7; ref/test_vectors.c:41,
 * ref/rng.c:155).  The library's own randombytes is a weak symbol, so a program
 * that defines the function overrides it the same way: with a counting
 * deterministic source, two runs of the single-call API give identical keys
 * and the bytes handed out are exactly 32 (d) + 32 (z) + 32 (m0) per
 * keypair+enc, in that order (ref/indcpa.c:215, ref/kem.c:32, ref/kem.c:59).
 */
#include <stdio.h>
#include <string.h>
#include "common.h"
#include "kyber_api.h"

static unsigned long calls, bytes;
static uint8_t state;
void randombytes(uint8_t *out, size_t n)
{
  calls++;
  bytes += n;
  while (n--) *out++ = state++;
}

int main(void)
{
  static uint8_t pk[2][pqcrystals_kyber768_PUBLICKEYBYTES], sk[2][pqcrystals_kyber768_SECRETKEYBYTES];
  static uint8_t ct[2][pqcrystals_kyber768_CIPHERTEXTBYTES], ss[2][32], back[32];
  int r;
  for (r = 0; r < 2; r++) {
    state = 7; calls = bytes = 0;
    if (pqcrystals_kyber768_hip_keypair(pk[r], sk[r]) || pqcrystals_kyber768_hip_enc(ct[r], ss[r], pk[r])) return 1;
    if (calls != 3 || bytes != 96) { printf("ERROR: %lu calls, %lu bytes\n", calls, bytes); return 1; }
  }
  if (memcmp(pk[0], pk[1], sizeof pk[0]) || memcmp(sk[0], sk[1], sizeof sk[0]) || memcmp(ct[0], ct[1], sizeof ct[0]) ||
      memcmp(ss[0], ss[1], 32)) { printf("ERROR: runs differ, the hook was not used\n"); return 1; }
  /* z is the tail of sk: bytes 39..70 of the counting source (d took 7..38) */
  if (sk[0][sizeof sk[0] - 32] != 39 || sk[0][sizeof sk[0] - 1] != 70) { printf("ERROR: z not from the hook\n"); return 1; }
  if (pqcrystals_kyber768_hip_dec(back, ct[0], sk[0]) || memcmp(back, ss[0], 32)) { printf("ERROR keys\n"); return 1; }
  printf("randombytes hook: OK\n");
  return 0;
}
