/*
 * mlkem_batch.h -- ML-KEM (FIPS 203) mode of libkyber_hip.so: batched
 * ML-KEM.KeyGen_internal / Encaps_internal / Decaps_internal on the MI355X.
 *
 * NO COUNTERPART IN THE REFERENCE: cothan/kyber implements round-3 Kyber only
 * (README.md:5; ref/kem.c:59-73 hashes m and applies a KDF over H(ct), which
 * the final standard dropped).  This mode is SURVEY.md row 8f-4, behind a
 * symbol set of its own so the reference-parity entry points of kyber_batch.h
 * never change.  PARITY UNPINNED: the reference holds no ML-KEM vectors and no
 * ML-KEM library is available offline; the mode is checked against two
 * restatements of FIPS 203 written for this repository (oracle/kyber_oracle.c
 * in C, tests/mlkem_model.py on hashlib), which pins nothing beyond their
 * shared reading of the standard.
 *
 * What is shared with the round-3 path, bit for bit (FIPS 203 algorithms
 * 13-15 = ref/indcpa.c:205-329): matrix expansion, CBD noise, NTT, base
 * multiplication, compression, encodings; object sizes and the layout of the
 * decapsulation key (dk_pke || ek || H(ek) || z = ref/params.h:65).
 * What differs (FIPS 203 algorithms 13, 17, 18):
 *   keygen  (rho, sigma) = G(d || k)            round 3: G(d)
 *   encaps  (K, r) = G(m || H(ek)), ss = K       round 3: m = H(m0), ss = KDF(K || H(ct))
 *   decaps  ss = K' or J(z || c)                 round 3: KDF(K' or z || H(ct))
 *
 * k = 2, 3, 4 selects ML-KEM-512 / -768 / -1024.  Dense arrays, instance i at
 * base + i * size; sizes are kyber_hip_{pk,sk,ct}_bytes(k).  coins are the
 * random inputs of the *_internal algorithms: d || z (64 bytes) for keygen,
 * m (32 bytes) for encaps; NULL draws them from randombytes().  Return 0 or a
 * negative KYBER_HIP_* code.  The *_dev forms take device pointers and a
 * hipStream_t like those of kyber_batch.h, under the same rules (16-byte
 * aligned pointers, else KYBER_HIP_EINVAL; calls may be captured in a graph).
 * Batches up to 65 536 take the small-batch routes of kyber_batch.h, on one
 * stream: during a decapsulation ss holds the rejection key J(z || c) until
 * the last kernel of the call has selected the secret.
 */
#ifndef MLKEM_BATCH_H
#define MLKEM_BATCH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int mlkem_batch_keypair(int k, size_t n, uint8_t *ek, uint8_t *dk, const uint8_t *coins /* n*64 or NULL */);
int mlkem_batch_encaps(int k, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *ek, const uint8_t *coins /* n*32 or NULL */);
int mlkem_batch_decaps(int k, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *dk);
int mlkem_batch_keypair_dev(int k, size_t n, uint8_t *ek, uint8_t *dk, const uint8_t *coins, void *stream);
int mlkem_batch_encaps_dev(int k, size_t n, uint8_t *ct, uint8_t *ss, const uint8_t *ek, const uint8_t *coins, void *stream);
int mlkem_batch_decaps_dev(int k, size_t n, uint8_t *ss, const uint8_t *ct, const uint8_t *dk, void *stream);

/* Input checks of FIPS 203 sections 7.2 and 7.3, one flag word per instance (0 = passes):
 *   bit 0  modulus check: some 12-bit coefficient of the encoded t^ is >= q
 *   bit 1  hash check (dk only): H(ek) differs from the hash stored in dk
 * (the type/length checks of the standard are the caller's: sizes are fixed here).
 * The encaps/decaps calls above are the *_internal algorithms and do not check. */
int mlkem_batch_check_ek(int k, size_t n, const uint8_t *ek, uint32_t *flags);
int mlkem_batch_check_dk(int k, size_t n, const uint8_t *dk, uint32_t *flags);
int mlkem_batch_check_dev(int k, size_t n, const uint8_t *key, int is_dk, uint32_t *flags, void *stream);

#ifdef __cplusplus
}
#endif
#endif
