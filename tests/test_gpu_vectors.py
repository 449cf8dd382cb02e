"""GPU: the reference's own golden gates, answered by the HIP engine.

* SHA256SUMS: the 10 000-iteration test_vectors stream (ref/test_vectors.c)
  rebuilt from ONE batched keypair/enc/dec of 10 000 instances on the GPU --
  no oracle involved, only the committed hashes.
* NIST KAT record 0 (ref/PQCgenKAT_kem.c) through the C-ABI.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import kyber_amd
import vectors

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = vectors.golden_sha256sums(os.path.join(HERE, "golden", "SHA256SUMS"))
META = json.load(open(os.path.join(HERE, "golden", "meta.json")))
BUILT = [name for name, (k, n90) in sorted(vectors.SETS.items())]


def _supported(k, n90):
    return kyber_amd.library().kyber_hip_supported(k, n90) == 1


@pytest.mark.parametrize("name", BUILT)
def test_gpu_reproduces_sha256sums(name):
    k, n90 = vectors.SETS[name]
    if not _supported(k, n90):
        pytest.skip("parameter set not built into libkyber_hip.so yet")
    n = 10000
    kc, ec = vectors.surf_coins(n)
    kem = kyber_amd.KyberHip(k, n90)
    pk, sk = kem.keypair(kc)
    ct, ssb = kem.enc(pk, ec)
    ssa = kem.dec(ct, sk)
    assert np.array_equal(ssa, ssb)
    assert vectors.tvec_sha256(kc, ec, pk, sk, ct, ssb, ssa) == GOLD[name]


@pytest.mark.parametrize("k,n90", sorted(vectors.META_NAME))
def test_gpu_nist_kat_record0(k, n90):
    if not _supported(k, n90):
        pytest.skip("parameter set not built into libkyber_hip.so yet")
    (seed, kc, m0), = vectors.kat_coins(1)
    kem = kyber_amd.KyberHip(k, n90)
    pk, sk = kem.keypair(np.frombuffer(kc, np.uint8))
    ct, ss = kem.enc(pk, np.frombuffer(m0, np.uint8))
    assert np.array_equal(kem.dec(ct, sk), ss)
    rec = vectors.kat_record(0, seed, pk[0], sk[0], ct[0], ss[0])
    assert hashlib.sha256(rec).hexdigest() == META[vectors.META_NAME[(k, n90)]]["nistkat-sha256"]
