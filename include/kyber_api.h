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

#define KYBER_API_DECLARE(SET, BASE)                                                   \
  enum {                                                                               \
    pqcrystals_##SET##_hip_SECRETKEYBYTES = pqcrystals_##BASE##_SECRETKEYBYTES,        \
    pqcrystals_##SET##_hip_PUBLICKEYBYTES = pqcrystals_##BASE##_PUBLICKEYBYTES,        \
    pqcrystals_##SET##_hip_CIPHERTEXTBYTES = pqcrystals_##BASE##_CIPHERTEXTBYTES,      \
    pqcrystals_##SET##_hip_BYTES = pqcrystals_##BASE##_BYTES                           \
  };                                                                                   \
  int pqcrystals_##SET##_hip_keypair(uint8_t *pk, uint8_t *sk);                        \
  int pqcrystals_##SET##_hip_enc(uint8_t *ct, uint8_t *ss, const uint8_t *pk);         \
  int pqcrystals_##SET##_hip_dec(uint8_t *ss, const uint8_t *ct, const uint8_t *sk);

KYBER_API_DECLARE(kyber512, kyber512)         /* ref/api.h:16-18 */
KYBER_API_DECLARE(kyber512_90s, kyber512)     /* ref/api.h:25-27 */
KYBER_API_DECLARE(kyber768, kyber768)         /* ref/api.h:39-41 */
KYBER_API_DECLARE(kyber768_90s, kyber768)     /* ref/api.h:48-50 */
KYBER_API_DECLARE(kyber1024, kyber1024)       /* ref/api.h:62-64 */
KYBER_API_DECLARE(kyber1024_90s, kyber1024)   /* ref/api.h:71-73 */

#ifdef __cplusplus
}
#endif
#endif
