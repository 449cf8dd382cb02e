"""tests/golden/kem.json holds digests the reference's own ref/ code produced
for seeded instances (tests/golden/make_kem_golden.py).  CPU: the oracle
reproduces them.  GPU: the library reproduces them -- this pins both to the
reference on a machine that has neither /root/reference nor oracle/_ref."""
import json
import os
import sys

import pytest

import vectors

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_kem_golden import kem_fixture  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "kem.json")))


class _OracleImpl:
    def __init__(self, o, k, n90):
        self.o, self.k, self.n90 = o, k, n90

    def keypair(self, c): return self.o.keypair(self.k, c, self.n90)
    def enc(self, pk, c): return self.o.enc(self.k, pk, c, self.n90)
    def dec(self, ct, sk): return self.o.dec(self.k, ct, sk, self.n90)


@pytest.mark.parametrize("k,n90", sorted(vectors.META_NAME))
def test_oracle_matches_reference_digest(oracle, k, n90):
    assert kem_fixture(_OracleImpl(oracle, k, n90), k, n90) == GOLD["sha256"][vectors.META_NAME[(k, n90)]]


@pytest.mark.gpu
@pytest.mark.parametrize("k,n90", sorted(vectors.META_NAME))
def test_gpu_matches_reference_digest(k, n90):
    import kyber_amd
    assert kem_fixture(kyber_amd.KyberHip(k, n90), k, n90) == GOLD["sha256"][vectors.META_NAME[(k, n90)]]
