"""CPU: the oracle's ML-KEM (FIPS 203) restatement against tests/mlkem_model.py,
an independent reading of the standard's numbered algorithms.  PARITY UNPINNED:
the reference has no ML-KEM code or vectors and none are available offline, so
agreement of the two shows a shared reading of the standard, nothing more.
Also pins what the mode shares with round-3 Kyber and where it must differ."""
import numpy as np
import pytest

import mlkem_model as mm


@pytest.mark.parametrize("k", [2, 3, 4])
def test_oracle_mlkem_matches_python_model(oracle, k):
    rng = np.random.default_rng(203 + k)
    n = 3
    kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    mc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ek, dk = oracle.mlkem_keypair(k, kc)
    ct, ss = oracle.mlkem_enc(k, ek, mc)
    ss2 = oracle.mlkem_dec(k, ct, dk)
    assert np.array_equal(ss, ss2)
    bad = ct.copy()
    bad[:, 5] ^= 0x10
    ssr = oracle.mlkem_dec(k, bad, dk)
    for i in range(n):
        mek, mdk = mm.keygen_internal(k, bytes(kc[i, :32]), bytes(kc[i, 32:]))
        assert mek == bytes(ek[i]) and mdk == bytes(dk[i])
        K, c = mm.encaps_internal(k, mek, bytes(mc[i]))
        assert c == bytes(ct[i]) and K == bytes(ss[i])
        assert mm.decaps_internal(k, mdk, c) == K
        assert mm.decaps_internal(k, mdk, bytes(bad[i])) == bytes(ssr[i]) != K
        assert mm.check_ek(k, mek)


def test_mlkem_differs_from_round3_only_in_the_fo_glue(oracle):
    """Same coins: the keys differ (G(d || k) vs G(d)), but encrypting under the SAME key
    with the same (m, r) gives the same ciphertext -- K-PKE is shared bit for bit."""
    k = 3
    rng = np.random.default_rng(7)
    kc = rng.integers(0, 256, (2, 64), dtype=np.uint8)
    mc = rng.integers(0, 256, (2, 32), dtype=np.uint8)
    pk3, sk3 = oracle.keypair(k, kc)
    ek, dk = oracle.mlkem_keypair(k, kc)
    assert not np.array_equal(pk3, ek)
    # round-3 encaps hashes the coins into m first: feed ML-KEM that m, against the same key
    import hashlib
    m = np.array([np.frombuffer(hashlib.sha3_256(bytes(c)).digest(), np.uint8) for c in mc])
    ct3, ss3 = oracle.enc(k, pk3, mc)
    ctm, ssm = oracle.mlkem_enc(k, pk3, m)
    assert np.array_equal(ct3, ctm) and not np.array_equal(ss3, ssm)
