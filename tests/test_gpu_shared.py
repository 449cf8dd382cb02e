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
