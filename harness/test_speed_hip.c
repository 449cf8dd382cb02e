/*
 * test_speed_hip -- throughput of the batched host-pointer entry points, the
 * counterpart of the reference's cycle-count benchmark (ref/test_speed.c:20-150,
 * which times one call at a time on one core).  For each batch size: median of
 * NRUNS wall-clock timings of keypair, enc and dec on pinned host buffers
 * (PCIe transfers included).  bench.py reports the device-resident rate.
 *
 *   test_speed_hip <512|768|1024>[-90s] [largest batch = 262144]
 */
#define _POSIX_C_SOURCE 199309L
#include <stdlib.h>
#include <time.h>
#include "common.h"
#include "kyber_batch.h"

#define NRUNS 7

static double now(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}
static int cmp(const void *a, const void *b) { return (*(const double *)a > *(const double *)b) - (*(const double *)a < *(const double *)b); }
static double median(double *v) { qsort(v, NRUNS, sizeof *v, cmp); return v[NRUNS / 2]; }

int main(int argc, char **argv)
{
  int k, n90;
  long nmax, n;
  if (parse_set(argc, argv, &k, &n90, &nmax, 262144) || nmax <= 0) {
    fprintf(stderr, "usage: %s <512|768|1024>[-90s] [largest batch]\n", argv[0]);
    return 2;
  }
  const size_t pkb = kyber_hip_pk_bytes(k), skb = kyber_hip_sk_bytes(k), ctb = kyber_hip_ct_bytes(k);
  uint8_t *pk = kyber_hip_host_alloc(nmax * pkb), *sk = kyber_hip_host_alloc(nmax * skb), *ct = kyber_hip_host_alloc(nmax * ctb);
  uint8_t *ss = kyber_hip_host_alloc(nmax * 32), *ss2 = kyber_hip_host_alloc(nmax * 32), *coins = kyber_hip_host_alloc(nmax * 64);
  if (!pk || !sk || !ct || !ss || !ss2 || !coins) { fprintf(stderr, "kyber_hip_host_alloc failed\n"); return 3; }
  randombytes(coins, nmax * 64);
  printf("Kyber%d%s, pinned host buffers, median of %d runs\n", 256 * k, n90 ? "-90s" : "", NRUNS);
  printf("%10s %14s %14s %14s %16s\n", "batch", "keypair/s", "enc/s", "dec/s", "us per enc call");
  for (n = 1; n <= nmax; n *= 16) {
    double tk[NRUNS], te[NRUNS], td[NRUNS];
    int r, rc = 0;
    for (r = -1; r < NRUNS; r++) { /* run -1 warms up */
      double t0 = now();
      rc |= kyber_batch_keypair(k, n90, n, pk, sk, coins);
      double t1 = now();
      rc |= kyber_batch_enc(k, n90, n, ct, ss, pk, coins);
      double t2 = now();
      rc |= kyber_batch_dec(k, n90, n, ss2, ct, sk);
      double t3 = now();
      if (rc || memcmp(ss, ss2, n * 32)) { fprintf(stderr, "failure at batch %ld (%s)\n", n, kyber_hip_last_error()); return 1; }
      if (r >= 0) { tk[r] = t1 - t0; te[r] = t2 - t1; td[r] = t3 - t2; }
    }
    const double mk = median(tk), me = median(te), md = median(td);
    printf("%10ld %14.0f %14.0f %14.0f %16.1f\n", n, n / mk, n / me, n / md, me * 1e6);
  }
  kyber_hip_host_free(pk); kyber_hip_host_free(sk); kyber_hip_host_free(ct);
  kyber_hip_host_free(ss); kyber_hip_host_free(ss2); kyber_hip_host_free(coins);
  return 0;
}
