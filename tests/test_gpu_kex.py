"""GPU key exchange (kyber_batch_kex_*) against the oracle and
the golden digests the reference itself produced (tests/golden/kex.json): the
reference's test_kex.c scenario, n instances at once, all six sets."""
import json
import os

import numpy as np
import pytest

import kex_scenario
import kyber_amd
import vectors
from kyber_amd.kex import Kex

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kex.json")))


@pytest.mark.parametrize("k,n90", sorted(vectors.META_NAME))
def test_kex_scenario_matches_reference_digest(k, n90):
    kem = kyber_amd.KyberHip(k, n90)
    res = kex_scenario.run(kem.keypair, Kex(k, n90), kex_scenario.coins_for(GOLD["n"], 4242 + 10 * k + n90))
    kex_scenario.check_agreement(res)
    assert kex_scenario.digest(res) == GOLD["sha256"][vectors.META_NAME[(k, n90)]]


@pytest.mark.parametrize("k,n90", [(2, 1), (3, 0), (4, 0)])
def test_kex_matches_oracle_field_by_field(oracle, k, n90):
    kem = kyber_amd.KyberHip(k, n90)
    c = kex_scenario.coins_for(150, 7 + k)
    got = kex_scenario.run(kem.keypair, Kex(k, n90), c)
    want = kex_scenario.run(lambda x: oracle.keypair(k, x, n90), oracle.kex(k, n90), c)
    for name in want:
        assert np.array_equal(got[name], want[name]), name


def test_kex_shared_static_key_of_b(oracle):
    """KYBER_KEX_SHARED_B: one server key for every instance gives the bytes of
    n runs with that key replicated."""
    k, n = 3, 40
    kem, kx = kyber_amd.KyberHip(k), Kex(k)
    c = kex_scenario.coins_for(n, 1234)
    pkb, skb = kem.keypair(c["kp_b"][:1])
    pka, ska = kem.keypair(c["kp_a"])
    okx = oracle.kex(k)
    rep_pk, rep_sk = np.tile(pkb, (n, 1)), np.tile(skb, (n, 1))
    senda, tk, eska = kx.ake_initA(pkb, c["ake_a"], shared_b=True)
    osenda, otk, oeska = okx.ake_initA(rep_pk, c["ake_a"])
    assert np.array_equal(senda, osenda) and np.array_equal(tk, otk) and np.array_equal(eska, oeska)
    sendb, kb = kx.ake_sharedB(senda, skb, pka, c["ake_b"], shared_b=True)
    osendb, okb = okx.ake_sharedB(osenda, rep_sk, pka, c["ake_b"])
    assert np.array_equal(sendb, osendb) and np.array_equal(kb, okb)
    assert np.array_equal(kx.ake_sharedA(sendb, tk, eska, ska), kb)
    usendb, ukb = kx.uake_sharedB(senda, skb, c["uake_b"], shared_b=True)
    ousendb, oukb = okx.uake_sharedB(osenda, rep_sk, c["uake_b"])
    assert np.array_equal(usendb, ousendb) and np.array_equal(ukb, oukb)


def test_kex_random_coins_and_errors():
    kx, kem = Kex(2), kyber_amd.KyberHip(2)
    pkb, skb = kem.keypair(n=3)
    senda, tk, eska = kx.uake_initA(pkb, n=3)          # coins from randombytes
    sendb, kb = kx.uake_sharedB(senda, skb)
    assert np.array_equal(kx.uake_sharedA(sendb, tk, eska), kb)
    with pytest.raises(ValueError):
        kx.uake_sharedB(senda, skb[:2])
    rc = kx.lib.kyber_batch_kex_uake_sharedA(7, 0, 1, None, None, None, None)
    assert rc == -1
