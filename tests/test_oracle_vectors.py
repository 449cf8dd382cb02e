"""CPU: pin the oracle to every golden vector the reference holds for the path:
all six SHA256SUMS lines (10 000 keypair/enc/dec each) and the six NIST KAT
record-0 hashes of the *_META.yml files."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
import vectors

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = vectors.golden_sha256sums(os.path.join(HERE, "golden", "SHA256SUMS"))
META = json.load(open(os.path.join(HERE, "golden", "meta.json")))


@pytest.mark.parametrize("name", sorted(vectors.SETS))
def test_oracle_tvec_binary_matches_sha256sums(name):
    """oracle/tvec prints the reference's test_vectors stream; 10 000 iterations."""
    k, n90 = vectors.SETS[name]
    exe = os.path.join(oracle_lib.ORACLE_DIR, "tvec")
    if not os.path.exists(exe):
        oracle_lib.build_oracle()
    p = subprocess.Popen([exe, str(k), str(n90), "10000"], stdout=subprocess.PIPE)
    h = hashlib.sha256()
    for chunk in iter(lambda: p.stdout.read(1 << 20), b""):
        h.update(chunk)
    assert p.wait() == 0
    assert h.hexdigest() == GOLD[name]
    assert h.hexdigest() == META[vectors.META_NAME[(k, n90)]]["testvectors-sha256"]


def test_python_stream_formatter_matches_tvec_binary(oracle):
    """The numpy formatter used by the GPU gate prints the same bytes as oracle/tvec."""
    n, k = 25, 3
    kc, ec = vectors.surf_coins(n)
    pk, sk = oracle.keypair(k, kc)
    ct, ss = oracle.enc(k, pk, ec)
    ss2 = oracle.dec(k, ct, sk)
    exe = os.path.join(oracle_lib.ORACLE_DIR, "tvec")
    ref = subprocess.run([exe, "3", "0", str(n)], stdout=subprocess.PIPE, check=True).stdout
    assert vectors.tvec_stream(kc, ec, pk, sk, ct, ss, ss2) == ref


@pytest.mark.parametrize("k,n90", sorted(vectors.META_NAME))
def test_oracle_nist_kat_record0(oracle, k, n90):
    """sha256 of PQCgenKAT_kem's record 0 == nistkat-sha256 of the META file."""
    (seed, kc, m0), = vectors.kat_coins(1)
    pk, sk = oracle.keypair(k, np.frombuffer(kc, np.uint8), n90)
    ct, ss = oracle.enc(k, pk, np.frombuffer(m0, np.uint8), n90)
    assert np.array_equal(oracle.dec(k, ct, sk, n90), ss)
    rec = vectors.kat_record(0, seed, pk[0], sk[0], ct[0], ss[0])
    assert hashlib.sha256(rec).hexdigest() == META[vectors.META_NAME[(k, n90)]]["nistkat-sha256"]
    assert (pk.shape[1], sk.shape[1], ct.shape[1]) == tuple(
        META[vectors.META_NAME[(k, n90)]][f] for f in ("length-public-key", "length-secret-key", "length-ciphertext"))


def test_oracle_symmetric_primitives_vs_hashlib(oracle):
    """SHA3/SHAKE/SHA-2 at the lengths the KEM uses (ref/kem.c:30,61-73; ref/symmetric.h)."""
    import ctypes
    rng = np.random.default_rng(1)
    lib = oracle.lib
    for n in (0, 1, 32, 33, 34, 64, 71, 72, 135, 136, 137, 768, 800, 1088, 1184, 1568):
        msg = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for fn, ref, outlen in (("ko_sha3_256", hashlib.sha3_256, 32), ("ko_sha3_512", hashlib.sha3_512, 64),
                                ("ko_sha256", hashlib.sha256, 32), ("ko_sha512", hashlib.sha512, 64)):
            out = ctypes.create_string_buffer(outlen)
            getattr(lib, fn)(out, msg, n)
            assert out.raw == ref(msg).digest(), (fn, n)
        for fn, ref in (("ko_shake128", hashlib.shake_128), ("ko_shake256", hashlib.shake_256)):
            for outlen in (32, 128, 168, 192, 504, 672):
                out = ctypes.create_string_buffer(outlen)
                getattr(lib, fn)(out, outlen, msg, n)
                assert out.raw == ref(msg).digest(outlen), (fn, n, outlen)


def test_oracle_aes256ctr_fips197_vector(oracle):
    """FIPS 197 appendix C.3: AES-256 of 00112233..ff under key 000102..1f."""
    import ctypes
    key = bytes(range(32))
    block = bytes.fromhex("00112233445566778899aabbccddeeff")
    out = ctypes.create_string_buffer(16)
    oracle.lib.ko_aes256ctr(out, 16, key, block[:12], int.from_bytes(block[12:], "big"))
    assert out.raw.hex() == "8ea2b7ca516745bfeafc49904b496089"
