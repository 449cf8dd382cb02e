"""Server-shaped entry points: n encapsulations to one public
key, n decapsulations with one secret key.  They must write exactly the bytes
of n independent reference calls (ref/kem.c:51-76, :94-127) -- checked against
the oracle on replicated keys, and against the per-key GPU entry points."""
import numpy as np
import pytest

import kyber_amd

pytestmark = pytest.mark.gpu


def _coins(n, width, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(n, width), dtype=np.uint8)


@pytest.mark.parametrize("n90", [0, 1])
@pytest.mark.parametrize("k", [2, 3, 4])
@pytest.mark.parametrize("n", [1, 9, 300])
def test_shared_key_matches_oracle(oracle, k, n, n90):
    kem = kyber_amd.KyberHip(k, n90)
    pk, sk = kem.keypair(_coins(1, 64, 900 + k))
    ec = _coins(n, 32, 901 + n)
    ct, ss = kem.enc_shared_pk(pk[0], ec)
    oct_, oss = oracle.enc(k, np.tile(pk, (n, 1)), ec, n90)
    assert np.array_equal(ct, oct_), "ct differs"
    assert np.array_equal(ss, oss), "ss differs"
    # a third of the ciphertexts corrupted: implicit rejection with the shared z
    bad = ct.copy()
    bad[::3, 5] ^= 0x40
    got = kem.dec_shared_sk(bad, sk[0])
    want = oracle.dec(k, bad, np.tile(sk, (n, 1)), n90)
    assert np.array_equal(got, want)
    keep = np.ones(n, bool)
    keep[::3] = False
    assert np.array_equal(got[keep], ss[keep])


def test_shared_key_large_batch_equals_per_key_path():
    """Several internal passes (chunk smaller than the batch), device tensors."""
    import torch
    from kyber_amd import lib as kl
    kem = kyber_amd.KyberHip(3)
    dev = torch.device("cuda", 0)
    n = 5000
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    kc = torch.randint(0, 256, (1, 64), dtype=torch.uint8, device=dev, generator=g)
    ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    pk, sk = kem.keypair_dev(kc)
    try:
        kl.set_chunk(1536)
        ct, ss = kem.enc_shared_pk_dev(pk[0], ec)
        ss2 = kem.dec_shared_sk_dev(ct, sk[0])
    finally:
        kl.set_chunk(0)
    ct_ref, ss_ref = kem.enc_dev(pk.expand(n, -1).contiguous(), ec)
    assert torch.equal(ct, ct_ref) and torch.equal(ss, ss_ref) and torch.equal(ss2, ss_ref)


@pytest.mark.parametrize("n90", [0, 1])
@pytest.mark.parametrize("k", [2, 3, 4])
def test_prepared_keys_match_oracle_and_shared_calls(oracle, k, n90):
    """kyber_hip_prepare_pk / _sk + kyber_batch_{enc,dec}_prepared: the expanded matrix and H(pk)
    come from the handle instead of being derived per call; bytes out are those of n reference
    calls with that key (ref/kem.c:51-76, :94-127), host and device forms, several calls per handle."""
    import torch
    kem = kyber_amd.KyberHip(k, n90)
    pk, sk = kem.keypair(_coins(1, 64, 700 + k))
    with kem.prepare_pk(pk[0]) as ppk, kem.prepare_sk(sk[0]) as psk:
        for n in (1, 7, 300):
            ec = _coins(n, 32, 701 + n)
            ct, ss = ppk.enc(ec)
            oct_, oss = oracle.enc(k, np.tile(pk, (n, 1)), ec, n90)
            assert np.array_equal(ct, oct_) and np.array_equal(ss, oss)
            ct2, ss2 = kem.enc_shared_pk(pk[0], ec)
            assert np.array_equal(ct, ct2) and np.array_equal(ss, ss2)
            bad = ct.copy()
            bad[::3, 11] ^= 0x01  # implicit rejection with the prepared key's z
            assert np.array_equal(psk.dec(bad), oracle.dec(k, bad, np.tile(sk, (n, 1)), n90))
            # device tensors
            dct, dss = ppk.enc_dev(torch.from_numpy(ec).cuda())
            assert np.array_equal(dct.cpu().numpy(), oct_) and np.array_equal(dss.cpu().numpy(), oss)
            assert np.array_equal(psk.dec_dev(dct).cpu().numpy(), oss)
    # handles prepared from device-resident keys
    with kem.prepare_pk(torch.from_numpy(pk[0]).cuda()) as ppk, kem.prepare_sk(torch.from_numpy(sk[0]).cuda()) as psk:
        ec = _coins(33, 32, 777)
        ct, ss = ppk.enc(ec)
        oct_, oss = oracle.enc(k, np.tile(pk, (33, 1)), ec, n90)
        assert np.array_equal(ct, oct_) and np.array_equal(ss, oss) and np.array_equal(psk.dec(ct), oss)


def test_prepared_key_misuse_is_refused():
    """a public-key handle cannot decapsulate, a secret-key handle cannot encapsulate, a freed or
    foreign pointer is refused: KYBER_HIP_EINVAL, nothing launched"""
    import ctypes
    kem = kyber_amd.KyberHip(3)
    pk, sk = kem.keypair(_coins(1, 64, 5))
    ppk, psk = kem.prepare_pk(pk[0]), kem.prepare_sk(sk[0])
    ct, ss = ppk.enc(_coins(4, 32, 6))
    with pytest.raises(kyber_amd.KyberHipError):
        ppk.dec(ct)
    with pytest.raises(kyber_amd.KyberHipError):
        psk.enc(_coins(4, 32, 6))
    out = np.zeros((4, 32), np.uint8)
    assert kem.lib.kyber_batch_dec_prepared(None, 4, out.ctypes.data_as(ctypes.c_void_p), ct.ctypes.data_as(ctypes.c_void_p)) == -1
    ppk.close()
    psk.close()
    ppk.close()  # idempotent
    # a large batch through a handle: several internal passes
    from kyber_amd import lib as kl
    with kem.prepare_pk(pk[0]) as p2, kem.prepare_sk(sk[0]) as s2:
        ec = _coins(5000, 32, 8)
        try:
            kl.set_chunk(1536)
            c, s = p2.enc(ec)
            assert np.array_equal(s2.dec(c), s)
        finally:
            kl.set_chunk(0)
        c0, s0 = kem.enc_shared_pk(pk[0], ec)
        assert np.array_equal(c, c0) and np.array_equal(s, s0)
