"""CPU: the oracle's key exchange (ko_kex_*, ref/kex.c restated) against the
golden digests computed by the reference itself (tests/golden/kex.json, made by
tests/golden/make_kex_golden.py) and, where oracle/_ref is built, against the
reference's ref/ and avx2/ code directly."""
import json
import os

import numpy as np
import pytest

import kex_scenario
import oracle_lib
import vectors

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kex.json")))


@pytest.mark.parametrize("k,n90", sorted(vectors.META_NAME))
def test_oracle_kex_matches_reference_digest(oracle, k, n90):
    res = kex_scenario.run(lambda c: oracle.keypair(k, c, n90), oracle.kex(k, n90),
                           kex_scenario.coins_for(GOLD["n"], 4242 + 10 * k + n90))
    kex_scenario.check_agreement(res)
    assert kex_scenario.digest(res) == GOLD["sha256"][vectors.META_NAME[(k, n90)]]
    pkb, skb, ctb = oracle.sizes(k)
    # ref/kex.h:7-12
    assert res["uake_senda"].shape[1] == pkb + ctb and res["uake_sendb"].shape[1] == ctb
    assert res["ake_senda"].shape[1] == pkb + ctb and res["ake_sendb"].shape[1] == 2 * ctb


@pytest.mark.parametrize("impl", ["ref", "avx2"])
@pytest.mark.parametrize("k,n90", [(2, 0), (3, 1), (4, 0)])
def test_oracle_kex_equals_reference(oracle, impl, k, n90):
    if impl == "avx2" and not oracle_lib.has_avx2():
        pytest.skip("host CPU lacks avx2/bmi2/aes")
    try:
        ref = oracle_lib.RefLib(impl, k, n90)
    except (OSError, FileNotFoundError):
        pytest.skip("oracle/_ref not built")
    c = kex_scenario.coins_for(12, 99 + k)
    a = kex_scenario.run(ref.keypair, ref.kex(), c)
    b = kex_scenario.run(lambda x: oracle.keypair(k, x, n90), oracle.kex(k, n90), c)
    for name in a:
        assert np.array_equal(a[name], b[name]), name


def test_wrong_static_key_breaks_agreement(oracle):
    """authentication property the scenario relies on: Bob answering with a
    different static key gives Alice a different session key."""
    k = 3
    c = kex_scenario.coins_for(4, 5)
    kx = oracle.kex(k)
    pkb, skb = oracle.keypair(k, c["kp_b"])
    _, sk_other = oracle.keypair(k, c["kp_a"])
    senda, tk, eska = kx.uake_initA(pkb, c["uake_a"])
    sendb, kb = kx.uake_sharedB(senda, sk_other, c["uake_b"])
    ka = kx.uake_sharedA(sendb, tk, eska)
    assert not np.any(np.all(ka == kb, axis=1))
