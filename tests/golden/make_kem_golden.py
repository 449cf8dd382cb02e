#!/usr/bin/env python3
"""Writes tests/golden/kem.json: per parameter set, sha256 over pk, sk, ct, ss
and the decapsulation results of half-corrupted ciphertexts for n = 64 seeded
instances, computed by THE REFERENCE ITSELF (oracle/_ref/libkyber_ref_*: the
reference's ref/ sources compiled in place).  Coins come from
numpy.random.default_rng(777 + 10 k + nineties); see kem_fixture() below, which
tests/test_golden_kem.py and the GPU tests reuse.

    python3 tests/golden/make_kem_golden.py      # needs /root/reference (build container only)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

N = 64


def kem_fixture(impl, k, n90):
    """impl offers keypair(coins), enc(pk, coins), dec(ct, sk) on numpy arrays -> hex digest"""
    rng = np.random.default_rng(777 + 10 * k + n90)
    kc = rng.integers(0, 256, (N, 64), dtype=np.uint8)
    ec = rng.integers(0, 256, (N, 32), dtype=np.uint8)
    pk, sk = impl.keypair(kc)
    ct, ss = impl.enc(pk, ec)
    bad = ct.copy()
    bad[::2, 11] ^= 0x80
    dec = impl.dec(bad, sk)
    h = hashlib.sha256()
    for a in (pk, sk, ct, ss, dec):
        h.update(np.ascontiguousarray(a, np.uint8).tobytes())
    return h.hexdigest()


if __name__ == "__main__":
    import oracle_lib
    out = {"n": N, "generator": "oracle/_ref libkyber_ref_* (reference ref/ sources)", "sha256": {}}
    for k in (2, 3, 4):
        for n90 in (0, 1):
            out["sha256"]["Kyber%d%s" % (256 * k, "-90s" if n90 else "")] = kem_fixture(oracle_lib.RefLib("ref", k, n90), k, n90)
    with open(os.path.join(HERE, "kem.json"), "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    print(json.dumps(out["sha256"], indent=1))
