/*
 * kyber_batch.h -- C-ABI batch entry points of libkyber_hip.so, the MI355X
 * (gfx950) engine for the round-3 Kyber KEM hot path.
 *
 * Plain C linkage, plain pointers and sizes.  Instance i of a batch lives at
 * base + i * <BYTES> in every array (dense, no padding), exactly the objects
 * the reference's three API functions take one at a time:
 *
 *   keypair  replaces  int crypto_kem_keypair(uint8_t *pk, uint8_t *sk)           ref/kem.c:23, ref/api.h:16,39,62
 *   enc      replaces  int crypto_kem_enc(uint8_t *ct, uint8_t *ss, const uint8_t *pk)   ref/kem.c:51, ref/api.h:17,40,63
 *   dec      replaces  int crypto_kem_dec(uint8_t *ss, const uint8_t *ct, const uint8_t *sk)  ref/kem.c:94, ref/api.h:18,41,64
 *
 * The reference draws its entropy through a link-time randombytes()
 * (ref/randombytes.h:7) in a fixed order: d (ref/indcpa.c:215), z
 * (ref/kem.c:32), m0 (ref/kem.c:59).  The batch calls take those draws as an
 * explicit `coins` array so that results are reproducible bit for bit;
 * coins == NULL makes the library draw them from randombytes() itself.
 *
 * Return value: 0 on success (the reference's functions always return 0,
 * ref/kem.c:33,74,126); negative on a HIP/runtime error or bad argument, the
 * only failure class this layer adds:
 *   KYBER_HIP_EINVAL (-1) bad parameter set or NULL pointer
 *   KYBER_HIP_ENOSYS (-2) parameter set not built into this library
 *   KYBER_HIP_EHIP   (-3) a HIP runtime call failed (kyber_hip_last_error())
 * Decapsulation failure is not an error: ss is then the pseudo-random value
 * the reference defines (implicit rejection, ref/kem.c:116-125).
 *
 * All sizes of one object are multiples of 16 bytes; device pointers passed
 * to the *_dev calls must be 16-byte aligned.
 */
#ifndef KYBER_BATCH_H
#define KYBER_BATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KYBER_HIP_EINVAL (-1)
#define KYBER_HIP_ENOSYS (-2)
#define KYBER_HIP_EHIP   (-3)

/* parameter-set selector for the generic entry points: k in {2,3,4}
 * (ref/params.h:4-8 KYBER_K), nineties = 1 for the -90s sets (KYBER_90S) */

/* 1 if the parameter set is built into this library, else 0 */
int kyber_hip_supported(int k, int nineties);

/* object sizes, = ref/params.h:63-66 / ref/api.h:6-9,29-32,52-55; 0 for a bad k */
size_t kyber_hip_pk_bytes(int k);
size_t kyber_hip_sk_bytes(int k);
size_t kyber_hip_ct_bytes(int k);
#define KYBER_HIP_SS_BYTES 32
#define KYBER_HIP_KEYPAIR_COINS 64 /* d || z  */
#define KYBER_HIP_ENC_COINS 32     /* m0      */

/* ---- host-memory batch calls (north_star's kyber_batch_{keypair,enc,dec}(n, ...)) ----
 * Buffers are ordinary host memory; the call copies in, runs on the current
 * HIP device and copies out before returning. */
int kyber_batch_keypair(int k, int nineties, size_t n, uint8_t *pk, uint8_t *sk, const uint8_t *coins /* n*64 or NULL */);
int kyber_batch_enc(int k, int nineties, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *pk, const uint8_t *coins /* n*32 or NULL */);
int kyber_batch_dec(int k, int nineties, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *sk);

/* ---- one node, several GPUs ----
 * The same three calls with the batch cut into `ndev` contiguous shards, shard g
 * on HIP device devices[g] (devices == NULL: devices 0 .. ndev-1), each driven by
 * a host thread of its own through the host-memory call above.  Instances are
 * independent (the reference's functions share no state, ref/kem.c:23-127), so no
 * device exchanges anything with another; results are byte for byte those of the
 * single-device call.  A device may be named more than once (its shards then
 * queue behind each other).  coins == NULL: all coins are drawn from randombytes()
 * by the calling thread, in instance order, before the shards start (an entropy
 * hook with static state, like ref/test_vectors.c:41, is never entered concurrently). */
int kyber_batch_keypair_mgpu(int k, int nineties, size_t n, uint8_t *pk, uint8_t *sk, const uint8_t *coins, int ndev, const int *devices);
int kyber_batch_enc_mgpu(int k, int nineties, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *pk, const uint8_t *coins, int ndev, const int *devices);
int kyber_batch_dec_mgpu(int k, int nineties, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *sk, int ndev, const int *devices);
/* The same for data that already lives on the GPUs: shard g is n[g] instances in device memory of
 * HIP device devices[g] (pk[g], sk[g], ... are that device's arrays; coins[g] must not be NULL), queued
 * on streams[g] (streams == NULL or streams[g] == NULL: that device's default stream).  Nothing is
 * copied, nothing is waited for: the call enqueues on each device in turn, restores the caller's
 * current device and returns; the devices run side by side.  Synchronise each stream (or device)
 * before reading results.  Returns the first error, its shard named in kyber_hip_last_error(). */
int kyber_batch_keypair_mgpu_dev(int k, int nineties, int ndev, const int *devices, const size_t *n, uint8_t *const *pk,
                                 uint8_t *const *sk, const uint8_t *const *coins, void *const *streams);
int kyber_batch_enc_mgpu_dev(int k, int nineties, int ndev, const int *devices, const size_t *n, uint8_t *const *ct, uint8_t *const *ss,
                             const uint8_t *const *pk, const uint8_t *const *coins, void *const *streams);
int kyber_batch_dec_mgpu_dev(int k, int nineties, int ndev, const int *devices, const size_t *n, uint8_t *const *ss,
                             const uint8_t *const *ct, const uint8_t *const *sk, void *const *streams);

/* ---- device-memory batch calls ----
 * Every pointer is device memory on the current HIP device (coins must not be
 * NULL); work is enqueued on `stream` (a hipStream_t passed as void*, NULL =
 * default stream) and the call returns without synchronising.  The internal
 * workspace is one per device and grows on demand; calls on one device -- from
 * any thread, on any stream -- are serialised by the library: a per-device lock
 * orders the launches and each call's stream first waits for an event recorded
 * behind the previous call's last kernel, so they may be issued concurrently
 * but do not run concurrently.  Encapsulations of up to 32768 instances of the
 * 90s sets (and unprepared shared-key calls) expand their matrices, and
 * decapsulations with one key for the whole batch hash their ciphertexts, on a
 * second, library-owned stream beside the rest of the call: it
 * starts behind everything already queued on
 * `stream` and `stream` waits for it before the arithmetic, so the caller sees
 * one ordered stream either way.  Under stream capture the call stays on
 * `stream` alone. */
int kyber_batch_keypair_dev(int k, int nineties, size_t n, uint8_t *pk, uint8_t *sk, const uint8_t *coins, void *stream);
int kyber_batch_enc_dev(int k, int nineties, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *pk, const uint8_t *coins, void *stream);
int kyber_batch_dec_dev(int k, int nineties, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *sk, void *stream);

/* ---- decapsulation that allocates nothing ----
 * No counterpart in the reference (whose functions never allocate anyway).  It
 * is for callers that cannot call hipMalloc: code under HIP graph capture,
 * tight serving loops, harnesses that forbid allocation.  Same arguments,
 * results and error codes as kyber_batch_dec_dev, but the internal workspace
 * is a fixed device array inside the library (about 40 MB per GPU, held until
 * the process ends) instead of a grow-on-demand allocation, so
 *   - n is limited to KYBER_HIP_NOALLOC_MAX_N instances per call; a larger n
 *     returns KYBER_HIP_EINVAL with a message in kyber_hip_last_error();
 *   - the call makes no hipMalloc / hipFree / allocating library call
 *     (with profiling off, see kyber_hip_profile_enable);
 *   - all calls on one GPU share that one array.  Calls made outside stream capture are
 *     ordered by the library like the workspace calls (each stream first waits for the
 *     previous call's last kernel).  Under capture nothing can be ordered for you: a
 *     captured call, or a graph replay, whose GPU work overlaps another no-allocation call
 *     on that GPU races on the array -- keep them on one stream or order them yourself;
 *   - the array keeps the last call's intermediates (m', r', Kbar', PRF bytes) until the
 *     next call overwrites them or kyber_hip_release() wipes them. */
#define KYBER_HIP_NOALLOC_MAX_N 4096
int kyber_batch_dec_dev_noalloc(int k, int nineties, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *sk, void *stream);
size_t kyber_hip_noalloc_max_n(void); /* = KYBER_HIP_NOALLOC_MAX_N of the library that is loaded */

/* ---- server-shaped variants: one key, many operations ----
 * enc_shared_pk: n encapsulations to ONE public key `pk` (KYBER_PUBLICKEYBYTES
 * bytes), e.g. n clients' worth of ciphertexts for one static server key.
 * dec_shared_sk: n ciphertexts decapsulated with ONE secret key `sk`.
 * Both write exactly what kyber_batch_enc / kyber_batch_dec write when given n
 * copies of the key (n x crypto_kem_enc / crypto_kem_dec, ref/kem.c:51,94);
 * internally H(pk) (ref/kem.c:64) and the matrix A^T (ref/indcpa.c:268-270)
 * are computed once instead of n times.  Same argument rules and error codes
 * as the per-key entry points; *_dev take device pointers and a hipStream_t. */
int kyber_batch_enc_shared_pk(int k, int nineties, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *pk, const uint8_t *coins);
int kyber_batch_dec_shared_sk(int k, int nineties, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *sk);
int kyber_batch_enc_shared_pk_dev(int k, int nineties, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *pk, const uint8_t *coins,
                                  void *stream);
int kyber_batch_dec_shared_sk_dev(int k, int nineties, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *sk, void *stream);

/* ---- prepared keys: the one-key calls without their per-call set-up ----
 * A shared-key call first expands the key's matrix (ref/indcpa.c:270: gen_matrix from rho) and, for
 * encapsulation, hashes the key (ref/kem.c:64: H(pk)) -- about 0.15 ms of single-key work that a small
 * batch (n <= 4096) feels on every call.  kyber_hip_prepare_pk / _sk do that once: the handle keeps a
 * copy of the key, its expanded matrix and H(pk) in device memory of the CURRENT device, and the
 * *_prepared calls are the shared-key calls minus that work -- same bytes out as
 * kyber_batch_enc_shared_pk / dec_shared_sk and as n calls of the reference with that key.
 * The reference has no such state (its functions re-derive everything per call, ref/kem.c:51-127);
 * this is SURVEY.md 8f-2's "precompute A^ and H(pk) once per pk".
 * prepare_* return after the handle is complete (they synchronise); a handle is read-only afterwards
 * and may serve any number of calls, from any thread, on any stream of its device.  _dev: the key is
 * device memory.  kyber_hip_prepared_free waits for the device, wipes a secret key's copy and frees. */
typedef struct kyber_hip_prepared kyber_hip_prepared;
int kyber_hip_prepare_pk(int k, int nineties, const uint8_t *pk, kyber_hip_prepared **out);
int kyber_hip_prepare_sk(int k, int nineties, const uint8_t *sk, kyber_hip_prepared **out);
int kyber_hip_prepare_pk_dev(int k, int nineties, const uint8_t *pk, kyber_hip_prepared **out);
int kyber_hip_prepare_sk_dev(int k, int nineties, const uint8_t *sk, kyber_hip_prepared **out);
int kyber_hip_prepared_free(kyber_hip_prepared *h);
int kyber_batch_enc_prepared(const kyber_hip_prepared *pk, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *coins /* n*32 or NULL */);
int kyber_batch_dec_prepared(const kyber_hip_prepared *sk, size_t n, uint8_t *ss, const uint8_t *ct);
int kyber_batch_enc_prepared_dev(const kyber_hip_prepared *pk, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *coins, void *stream);
int kyber_batch_dec_prepared_dev(const kyber_hip_prepared *sk, size_t n, uint8_t *ss, const uint8_t *ct, void *stream);

/* ---- key exchange: ref/kex.c:6-54, ref/kex.h:7-29, batched ----
 * Each call is n independent runs of the reference function of the same name.
 * Message layouts per instance are the reference's (ref/kex.h:7-12):
 *   UAKE/AKE send A = pk || ct           (PUBLICKEYBYTES + CIPHERTEXTBYTES)
 *   UAKE send B     = ct                 (CIPHERTEXTBYTES)
 *   AKE  send B     = ct || ct'          (2 x CIPHERTEXTBYTES)
 *   tk, key         = 32 bytes           (KEX_SSBYTES)
 * all dense, instance i at base + i*size.  coins are the randombytes draws of
 * the KEM calls inside, in call order: initA d||z||m0 (96 B per instance),
 * uake_sharedB m0 (32), ake_sharedB m0||m0' (64); NULL = draw from randombytes
 * (host-pointer forms only).  flags: KYBER_KEX_SHARED_B = B's static key (pkb
 * in initA, skb in sharedB) is ONE key used for every instance, not n keys.
 * Return 0 or a negative KYBER_HIP_* code (the reference returns void). */
#define KYBER_KEX_SHARED_B 1
int kyber_batch_kex_uake_initA(int k, int nineties, size_t n, uint8_t *send, uint8_t *tk, uint8_t *sk, const uint8_t *pkb,
                               const uint8_t *coins, int flags);
int kyber_batch_kex_uake_sharedB(int k, int nineties, size_t n, uint8_t *send, uint8_t *key, const uint8_t *recv,
                                 const uint8_t *skb, const uint8_t *coins, int flags);
int kyber_batch_kex_uake_sharedA(int k, int nineties, size_t n, uint8_t *key, const uint8_t *recv, const uint8_t *tk,
                                 const uint8_t *sk);
int kyber_batch_kex_ake_initA(int k, int nineties, size_t n, uint8_t *send, uint8_t *tk, uint8_t *sk, const uint8_t *pkb,
                              const uint8_t *coins, int flags);
int kyber_batch_kex_ake_sharedB(int k, int nineties, size_t n, uint8_t *send, uint8_t *key, const uint8_t *recv,
                                const uint8_t *skb, const uint8_t *pka, const uint8_t *coins, int flags);
int kyber_batch_kex_ake_sharedA(int k, int nineties, size_t n, uint8_t *key, const uint8_t *recv, const uint8_t *tk,
                                const uint8_t *sk, const uint8_t *ska);
/* device pointers, work queued on `stream` (temporaries are stream-ordered allocations) */
int kyber_batch_kex_uake_initA_dev(int k, int nineties, size_t n, uint8_t *send, uint8_t *tk, uint8_t *sk, const uint8_t *pkb,
                                   const uint8_t *coins, int flags, void *stream);
int kyber_batch_kex_uake_sharedB_dev(int k, int nineties, size_t n, uint8_t *send, uint8_t *key, const uint8_t *recv,
                                     const uint8_t *skb, const uint8_t *coins, int flags, void *stream);
int kyber_batch_kex_uake_sharedA_dev(int k, int nineties, size_t n, uint8_t *key, const uint8_t *recv, const uint8_t *tk,
                                     const uint8_t *sk, void *stream);
int kyber_batch_kex_ake_initA_dev(int k, int nineties, size_t n, uint8_t *send, uint8_t *tk, uint8_t *sk, const uint8_t *pkb,
                                  const uint8_t *coins, int flags, void *stream);
int kyber_batch_kex_ake_sharedB_dev(int k, int nineties, size_t n, uint8_t *send, uint8_t *key, const uint8_t *recv,
                                    const uint8_t *skb, const uint8_t *pka, const uint8_t *coins, int flags, void *stream);
int kyber_batch_kex_ake_sharedA_dev(int k, int nineties, size_t n, uint8_t *key, const uint8_t *recv, const uint8_t *tk,
                                    const uint8_t *sk, const uint8_t *ska, void *stream);

/* per-set spellings, e.g. kyber768_batch_enc(n, ct, ss, pk, coins) */
#define KYBER_HIP_DECLARE_SET(NAME)                                                                         \
  int NAME##_batch_keypair(size_t n, uint8_t *pk, uint8_t *sk, const uint8_t *coins);                       \
  int NAME##_batch_enc(size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *pk, const uint8_t *coins);        \
  int NAME##_batch_dec(size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *sk);
KYBER_HIP_DECLARE_SET(kyber512)
KYBER_HIP_DECLARE_SET(kyber768)
KYBER_HIP_DECLARE_SET(kyber1024)
KYBER_HIP_DECLARE_SET(kyber512_90s)
KYBER_HIP_DECLARE_SET(kyber768_90s)
KYBER_HIP_DECLARE_SET(kyber1024_90s)

/* ---- batched polynomial arithmetic on device memory (BASELINE.json config 2) ----
 * polys are dense int16_t[256] records (the reference's `poly`, ref/poly.h:11-13).
 *   ntt        in place, = poly_ntt            ref/poly.c:244-248 (ntt + Barrett)
 *   invntt     in place, = poly_invntt_tomont  ref/poly.c:259-262
 *   basemul    r = a o b, = poly_basemul_montgomery  ref/poly.c:273-280
 *   mul_fused  r = invntt_tomont(poly_reduce(poly_ntt(a) o poly_ntt(b))) in one kernel
 * r may be the same array as a or b (polynomial i of r depends on polynomial i
 * of a and b only); partially overlapping arrays are not supported. */
int kyber_batch_poly_ntt_dev(size_t npoly, int16_t *r, void *stream);
int kyber_batch_poly_invntt_dev(size_t npoly, int16_t *r, void *stream);
int kyber_batch_poly_basemul_dev(size_t npoly, int16_t *r, const int16_t *a, const int16_t *b, void *stream);
int kyber_batch_poly_mul_fused_dev(size_t npoly, int16_t *r, const int16_t *a, const int16_t *b, void *stream);
/* The same four with CANONICAL output: every coefficient is the representative in [0, q) of what the
 * reference function leaves in its int16 -- the value its own serializers make of it before any byte
 * is written (ref/poly.c:118-121, ref/polyvec.c:24-26), so nothing downstream can tell the difference.
 * These are the fast path (exact-integer fp32 arithmetic, kyber_amd/csrc/wpoly.h); the four above
 * return the reference's exact representatives (invntt, basemul and mul_fused at about half the rate).
 *   ntt_canon        = poly_ntt mod q                                  ref/poly.c:244-248, ref/ntt.c:80-95
 *   invntt_canon     = poly_invntt_tomont mod q (factor 2^16 included) ref/poly.c:259-262, ref/ntt.c:106-126
 *   basemul_canon    = poly_basemul_montgomery mod q (factor 2^-16)    ref/poly.c:273-280, ref/ntt.c:139-146
 *   mul_fused_canon  = invntt_tomont(poly_ntt(a) o poly_ntt(b)) mod q = a * b in Z_q[X]/(X^256 + 1)
 * Inputs: any int16 (values beyond what the reference's int16 arithmetic could take are simply
 * reduced mod q first). */
int kyber_batch_poly_ntt_canon_dev(size_t npoly, int16_t *r, void *stream);
int kyber_batch_poly_invntt_canon_dev(size_t npoly, int16_t *r, void *stream);
int kyber_batch_poly_basemul_canon_dev(size_t npoly, int16_t *r, const int16_t *a, const int16_t *b, void *stream);
int kyber_batch_poly_mul_fused_canon_dev(size_t npoly, int16_t *r, const int16_t *a, const int16_t *b, void *stream);

/* ---- stage-level device entry points (parity tests of single kernels) ----
 * gen_matrix: mat[i][e][256] int16, e = row*k + col of A (transposed = 0) or
 * A^T (transposed = 1) expanded from seeds[i][32].        ref/indcpa.c:164-192 */
int kyber_batch_gen_matrix_dev(int k, int nineties, size_t n, int16_t *mat, const uint8_t *seeds, int transposed, void *stream);
/* keccak_f1600: the permutation on n states of 25 little-endian 64-bit words, in place.   ref/fips202.c:82-344
 * hash_h: out[i][32] = H(in[i]) with in[i] one public key (of_ct = 0) or one ciphertext (of_ct = 1) of
 * the set: sha3_256, or sha256 for the 90s sets (ref/symmetric.h:27,50), through the FO kernels' absorb loops */
int kyber_batch_keccak_f1600_dev(size_t n, uint64_t *states, void *stream);
int kyber_batch_hash_h_dev(int k, int nineties, size_t n, int of_ct, uint8_t *out, const uint8_t *in, void *stream);
/* hash_g: out[i][64] = G(in[i]) with in[i] of inlen = 32 bytes (G(d), ref/indcpa.c:216) or 64 bytes
 * (G(m || H(pk)), ref/kem.c:65,111): sha3_512, or sha512 for the 90s sets (ref/symmetric.h:28,51;
 * ref/fips202.c:765-774, ref/sha512.c:77-302), through the device functions of the FO kernels */
int kyber_batch_hash_g_dev(int nineties, size_t n, size_t inlen, uint8_t *out, const uint8_t *in, void *stream);
/* aes256ctr: out[i][nblocks][16] = keystream blocks first_block .. first_block + nblocks - 1 of
 * AES-256-CTR(keys[i][32], nonce = nonces[i] as two little-endian bytes followed by ten zero bytes), the
 * XOF and PRF of the 90s sets: aes256ctr_init + aes256ctr_squeezeblocks / aes256ctr_prf, whose callers
 * pass exactly such nonces.                    ref/aes256ctr.c:530-586, ref/symmetric-aes.c:7-20 */
int kyber_batch_aes256ctr_dev(size_t n, uint8_t *out, size_t nblocks, const uint8_t *keys, const uint16_t *nonces,
                              uint32_t first_block, void *stream);
/* getnoise: r[i][nonce][256] = poly_getnoise(seeds[i], nonce) for nonce = 0..nnonce-1, with eta1 for
 * nonce < n_eta1 and eta2 above -- the PRF kernel of the KEM pipelines followed by the centred
 * binomial map.                      ref/poly.c:208-232, ref/symmetric-shake.c:43-51 / symmetric-aes.c:15-20
 * cbd: r[i] = cbd_eta(buf[i]), buf = 64*eta bytes per polynomial, eta = 2 or 3.        ref/cbd.c:58-107 */
int kyber_batch_getnoise_dev(int k, int nineties, size_t n, int nnonce, int n_eta1, int16_t *r, const uint8_t *seeds, void *stream);
int kyber_batch_poly_cbd_dev(int eta, size_t npoly, int16_t *r, const uint8_t *buf, void *stream);
/* codecs, polynomial by polynomial, through the device functions the KEM kernels use:
 *   tobytes / frombytes   384 bytes per polynomial                          ref/poly.c:111-145
 *   compress / decompress 32*d bytes per polynomial, d = 4, 5 (poly_compress, ref/poly.c:18-100),
 *                         d = 10, 11 (one polynomial of polyvec_compress, ref/polyvec.c:15-115),
 *                         d = 1 (poly_tomsg / poly_frommsg, ref/poly.c:155-194)
 * compress and tobytes accept any representative in (-q, q) (|a| <= 5000 for compress). */
int kyber_batch_poly_tobytes_dev(size_t npoly, uint8_t *r, const int16_t *a, void *stream);
int kyber_batch_poly_frombytes_dev(size_t npoly, int16_t *r, const uint8_t *a, void *stream);
int kyber_batch_poly_compress_dev(int d, size_t npoly, uint8_t *r, const int16_t *a, void *stream);
int kyber_batch_poly_decompress_dev(int d, size_t npoly, int16_t *r, const uint8_t *a, void *stream);
/* polyvec_basemul_acc on the fp32 engine, as the KEM kernels accumulate a matrix row against the
 * cached multiplicand: r[i] = centred residue of sum_{j<k} a[i][j] o b[i][j] WITHOUT the 2^-16 of the
 * reference's Montgomery products (ref/polyvec.c:189-201 gives r * 2^-16 mod q); a[i][j] canonical
 * in [0, q), b[i][j] centred with |b| <= 1665. */
int kyber_batch_polyvec_basemul_acc_dev(int k, size_t nvec, int16_t *r, const int16_t *a, const int16_t *b, void *stream);

/* ---- pinned host memory (optional) ----
 * The host-pointer entry points overlap PCIe copies with the kernels; the
 * copies are true DMA only from page-locked memory.  Buffers from
 * kyber_hip_host_alloc (hipHostMalloc) give the full rate, any other host
 * memory works at the runtime's staged-copy rate.  No counterpart in the
 * reference, whose buffers are plain arrays. */
void *kyber_hip_host_alloc(size_t bytes);
void kyber_hip_host_free(void *p);

/* ---- housekeeping ---- */
/* instances processed per internal pass (workspace sizing); 0 restores the default
 * (2^19), more than 2^24 is refused with KYBER_HIP_EINVAL */
int kyber_hip_set_chunk(size_t instances);
/* Workspaces per device, 1 (the default) to 8.  No counterpart in the reference, whose calls
 * share nothing.  With one workspace the GPU work of any two pipeline calls on a device is
 * ordered, whatever streams they are queued on.  With s > 1, calls of up to 65 536 instances
 * that arrive on different streams, or through the host-pointer entry points from different
 * threads (e.g. the request threads of a server), take different workspaces and run side by
 * side on the device, which such batches leave mostly idle.  In this mode no call uses the
 * library's second stream and a workspace that changes hands between streams is waited for on
 * the host, so no stream ever waits on the device for another.  Calls on one stream and larger
 * batches behave as before; a call under stream capture takes the workspace that a call of its
 * size outside capture laid out last (it cannot allocate), whichever that is.  Each extra workspace is sized by
 * the largest batch it has served (about 5.7 KB per Kyber768 instance).  Takes effect for the
 * calls that start after it; use at least as many slots as there are calling threads. */
int kyber_hip_set_workspace_slots(int slots);
/* Tuning/test aid (no counterpart in the reference): upper bound on the records
 * of the matrix sampler's tail queue per pass.  Entries that do not fit are
 * finished by the main kernel itself, so any value gives the same output;
 * (size_t)-1 restores the default (1/16 of the entries + 64). */
int kyber_hip_set_tail_capacity(size_t records);
/* Release what the library holds on EVERY device it has run on: workspace, host-path staging
 * buffers and streams are wiped and freed, the no-allocation scratch array is wiped (it is part of the
 * code object and stays).  Waits for each such device to be idle first.  The next call rebuilds
 * what it needs.  Prepared-key handles are the caller's (kyber_hip_prepared_free). */
int kyber_hip_release(void);
/* text of the last HIP error seen by this thread ("" if none) */
const char *kyber_hip_last_error(void);
/* Measurement aid (no counterpart in the reference): with profiling on, every
 * kernel launch of the *_dev calls is bracketed by HIP events on its stream.
 * kyber_hip_profile_read synchronises the device and writes one text line per
 * kernel name, "<name> <launches> <total_ms>\n", into buf. */
int kyber_hip_profile_enable(int on);
int kyber_hip_profile_read(char *buf, size_t len);

/* Entropy hook, same contract as the reference's (ref/randombytes.h:7).  The
 * library ships a weak default backed by getrandom(2); a program that defines
 * its own randombytes() overrides it at link time, as the reference's test
 * harnesses do (ref/test_vectors.c:41). */
void randombytes(uint8_t *out, size_t outlen);

#ifdef __cplusplus
}
#endif
#endif
