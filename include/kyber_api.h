/*
 * kyber_api.h -- the reference's namespaced single-call KEM API, served by
 * the HIP engine.  Same names and size macros as ref/api.h with the
 * implementation tag `hip` in place of `ref`/`avx2`/`neon`
 * (ref/params.h:11-31 KYBER_NAMESPACE, ref/kem.h:32-38), same argument order,
 * same "always returns 0" contract unless the GPU runtime itself fails.
 *
 * Each call is a batch of one on the current HIP device; entropy comes from
 * randombytes() in the reference's order: 32 bytes d, 32 bytes z for keypair
 * (ref/indcpa.c:215, ref/kem.c:32), 32 bytes m0 for enc (ref/kem.c:59).
 */
#ifndef KYBER_API_H
#define KYBER_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define pqcrystals_kyber512_SECRETKEYBYTES 1632
#define pqcrystals_kyber512_PUBLICKEYBYTES 800
#define pqcrystals_kyber512_CIPHERTEXTBYTES 768
#define pqcrystals_kyber512_BYTES 32

#define pqcrystals_kyber768_SECRETKEYBYTES 2400
#define pqcrystals_kyber768_PUBLICKEYBYTES 1184
#define pqcrystals_kyber768_CIPHERTEXTBYTES 1088
#define pqcrystals_kyber768_BYTES 32

#define pqcrystals_kyber1024_SECRETKEYBYTES 3168
#define pqcrystals_kyber1024_PUBLICKEYBYTES 1568
#define pqcrystals_kyber1024_CIPHERTEXTBYTES 1568
#define pqcrystals_kyber1024_BYTES 32

/* Sizes per implementation-tagged name: preprocessor macros, as in the reference, so that
 * `#if pqcrystals_kyber768_hip_CIPHERTEXTBYTES != 1088` style checks written against ref/api.h
 * compile against this header too. */
/* ref/api.h:11-18 */
#define pqcrystals_kyber512_hip_SECRETKEYBYTES pqcrystals_kyber512_SECRETKEYBYTES
#define pqcrystals_kyber512_hip_PUBLICKEYBYTES pqcrystals_kyber512_PUBLICKEYBYTES
#define pqcrystals_kyber512_hip_CIPHERTEXTBYTES pqcrystals_kyber512_CIPHERTEXTBYTES
#define pqcrystals_kyber512_hip_BYTES pqcrystals_kyber512_BYTES
/* ref/api.h:20-27 */
#define pqcrystals_kyber512_90s_hip_SECRETKEYBYTES pqcrystals_kyber512_SECRETKEYBYTES
#define pqcrystals_kyber512_90s_hip_PUBLICKEYBYTES pqcrystals_kyber512_PUBLICKEYBYTES
#define pqcrystals_kyber512_90s_hip_CIPHERTEXTBYTES pqcrystals_kyber512_CIPHERTEXTBYTES
#define pqcrystals_kyber512_90s_hip_BYTES pqcrystals_kyber512_BYTES
/* ref/api.h:34-41 */
#define pqcrystals_kyber768_hip_SECRETKEYBYTES pqcrystals_kyber768_SECRETKEYBYTES
#define pqcrystals_kyber768_hip_PUBLICKEYBYTES pqcrystals_kyber768_PUBLICKEYBYTES
#define pqcrystals_kyber768_hip_CIPHERTEXTBYTES pqcrystals_kyber768_CIPHERTEXTBYTES
#define pqcrystals_kyber768_hip_BYTES pqcrystals_kyber768_BYTES
/* ref/api.h:43-50 */
#define pqcrystals_kyber768_90s_hip_SECRETKEYBYTES pqcrystals_kyber768_SECRETKEYBYTES
#define pqcrystals_kyber768_90s_hip_PUBLICKEYBYTES pqcrystals_kyber768_PUBLICKEYBYTES
#define pqcrystals_kyber768_90s_hip_CIPHERTEXTBYTES pqcrystals_kyber768_CIPHERTEXTBYTES
#define pqcrystals_kyber768_90s_hip_BYTES pqcrystals_kyber768_BYTES
/* ref/api.h:57-64 */
#define pqcrystals_kyber1024_hip_SECRETKEYBYTES pqcrystals_kyber1024_SECRETKEYBYTES
#define pqcrystals_kyber1024_hip_PUBLICKEYBYTES pqcrystals_kyber1024_PUBLICKEYBYTES
#define pqcrystals_kyber1024_hip_CIPHERTEXTBYTES pqcrystals_kyber1024_CIPHERTEXTBYTES
#define pqcrystals_kyber1024_hip_BYTES pqcrystals_kyber1024_BYTES
/* ref/api.h:66-73 */
#define pqcrystals_kyber1024_90s_hip_SECRETKEYBYTES pqcrystals_kyber1024_SECRETKEYBYTES
#define pqcrystals_kyber1024_90s_hip_PUBLICKEYBYTES pqcrystals_kyber1024_PUBLICKEYBYTES
#define pqcrystals_kyber1024_90s_hip_CIPHERTEXTBYTES pqcrystals_kyber1024_CIPHERTEXTBYTES
#define pqcrystals_kyber1024_90s_hip_BYTES pqcrystals_kyber1024_BYTES

#define KYBER_API_DECLARE(SET)                                                         \
  int pqcrystals_##SET##_hip_keypair(uint8_t *pk, uint8_t *sk);                        \
  int pqcrystals_##SET##_hip_enc(uint8_t *ct, uint8_t *ss, const uint8_t *pk);         \
  int pqcrystals_##SET##_hip_dec(uint8_t *ss, const uint8_t *ct, const uint8_t *sk);

KYBER_API_DECLARE(kyber512)
KYBER_API_DECLARE(kyber512_90s)
KYBER_API_DECLARE(kyber768)
KYBER_API_DECLARE(kyber768_90s)
KYBER_API_DECLARE(kyber1024)
KYBER_API_DECLARE(kyber1024_90s)

#ifdef __cplusplus
}
#endif
#endif
