"""harness/: the reference's vector programs (ref/test_vectors.c,
ref/PQCgenKAT_kem.c) re-linked against the GPU library.

CPU: the binaries exist and their own PRNGs (surf, AES-256 CTR_DRBG) hand out
the same coins as the test-side implementations that the oracle gates use.
GPU: harness/check.sh == the reference's `sha256sum -c SHA256SUMS` gate plus
KAT record 0 for all six parameter sets."""
import os
import subprocess

import numpy as np
import pytest

import vectors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = os.path.join(ROOT, "harness")


def _run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, timeout=600).stdout


def test_surf_coins_match():
    n = 40
    lines = _run(os.path.join(H, "test_vectors_hip"), "768", str(n), "--coins").split()
    kc, ec = vectors.surf_coins(n)
    assert len(lines) == 2 * n
    for i in range(n):
        assert lines[2 * i].decode() == bytes(kc[i]).hex()
        assert lines[2 * i + 1].decode() == bytes(ec[i]).hex()


def test_drbg_coins_match():
    n = 5
    lines = _run(os.path.join(H, "kat_hip"), "1024-90s", str(n), "--coins").split()
    want = vectors.kat_coins(n)
    assert len(lines) == 3 * n
    for i, (seed, kc, m0) in enumerate(want):
        assert [l.decode() for l in lines[3 * i:3 * i + 3]] == [seed.hex(), kc.hex(), m0.hex()]


def test_usage_errors():
    for exe in ("test_vectors_hip", "kat_hip", "test_kyber_hip", "test_kex_hip", "test_speed_hip"):
        r = subprocess.run([os.path.join(H, exe), "999"], capture_output=True)
        assert r.returncode == 2 and b"usage" in r.stderr


@pytest.mark.gpu
def test_check_sh_all_sets():
    r = subprocess.run([os.path.join(H, "check.sh")], capture_output=True, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0, out + r.stderr.decode()
    assert out.count(": OK") == 12 and "FAILED" not in out


@pytest.mark.gpu
def test_kat_rsp_format():
    """Full .rsp layout (header, blank line after every record) and agreement
    with the Python binding on the same DRBG coins."""
    import kyber_amd
    out = _run(os.path.join(H, "kat_hip"), "512", "3").decode()
    assert out.startswith("# Kyber512\n\ncount = 0\nseed = ")
    recs = out.split("\n\n")[1:]
    assert recs[-1] == "" and len(recs) == 4
    kem = kyber_amd.KyberHip(2)
    coins = vectors.kat_coins(3)
    kc = np.frombuffer(b"".join(c[1] for c in coins), np.uint8).reshape(3, 64)
    ec = np.frombuffer(b"".join(c[2] for c in coins), np.uint8).reshape(3, 32)
    pk, sk = kem.keypair(kc)
    ct, ss = kem.enc(pk, ec)
    for i in range(3):
        assert recs[i] + "\n" == vectors.kat_record(i, coins[i][0], pk[i], sk[i], ct[i], ss[i]).decode()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["512", "768", "1024", "512-90s", "768-90s", "1024-90s"])
def test_reference_functional_tests_on_gpu(name):
    """ref/test_kyber.c (1000 rounds of keys / invalid sk / invalid ciphertext)
    and ref/test_kex.c relinked against the library: same pass criteria, same
    size lines on stdout (ref/test_kyber.c:107-109, ref/test_kex.c:60-63)."""
    k = {"512": 2, "768": 3, "1024": 4}[name.split("-")[0]]
    pkb, skb, ctb = 384 * k + 32, 768 * k + 96, 32 * ((11 if k == 4 else 10) * k + (5 if k == 4 else 4))
    out = _run(os.path.join(H, "test_kyber_hip"), name).decode()
    assert out == "CRYPTO_SECRETKEYBYTES:  %d\nCRYPTO_PUBLICKEYBYTES:  %d\nCRYPTO_CIPHERTEXTBYTES: %d\n" % (skb, pkb, ctb)
    out = _run(os.path.join(H, "test_kex_hip"), name).decode()
    assert out == ("KEX_UAKE_SENDABYTES: %d\nKEX_UAKE_SENDBBYTES: %d\nKEX_AKE_SENDABYTES: %d\nKEX_AKE_SENDBBYTES: %d\n"
                   % (pkb + ctb, ctb, pkb + ctb, 2 * ctb))


@pytest.mark.gpu
def test_link_time_randombytes_hook():
    """a program's own randombytes replaces the library's weak one, as the
    reference's harnesses do (ref/randombytes.h:7, ref/test_vectors.c:41)"""
    assert _run(os.path.join(H, "test_randombytes_hook")).decode() == "randombytes hook: OK\n"
