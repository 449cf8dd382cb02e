#!/usr/bin/env python3
"""Soak test of the ML-KEM mode (include/mlkem_batch.h): GPU against the oracle's restatement of FIPS 203
(oracle/kyber_oracle.c; PARITY UNPINNED, see its header) on batches of random size, every byte compared --
key generation, encapsulation, decapsulation of valid, corrupted and wrong-key ciphertexts.
Test infrastructure (uses oracle_lib.py beside it), run by hand, not collected by pytest.

    python3 tests/soak_mlkem.py [seconds per set = 60] [largest batch = 90000] [seed shift = 0]
"""
import hashlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # this directory
import kyber_amd  # noqa: E402
import oracle_lib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 90000
shift = int(sys.argv[3]) if len(sys.argv) > 3 else 0
threads = 16
with open(kyber_amd.lib.library_path(), "rb") as f:
    print("library sha256 %s, seed shift %d" % (hashlib.sha256(f.read()).hexdigest()[:12], shift), flush=True)
oracle = oracle_lib.load_oracle()
for k in (3, 2, 4):
    ml = kyber_amd.MlKemHip(k)
    seed = 2_000_003 * k + 7919 * shift
    rng = np.random.default_rng(seed)
    t0, total, rounds = time.time(), 0, 0
    while time.time() - t0 < budget:
        n = int(rng.integers(1, nmax + 1))
        kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
        mc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        ek, dk = ml.keypair(kc)
        ct, ss = ml.encaps(ek, mc)
        bad = ct.copy()
        pos = rng.integers(0, ct.shape[1], size=n)
        bad[np.arange(n), pos] ^= rng.integers(1, 256, size=n, dtype=np.uint8)
        bad[::2] = ct[::2]                       # every other ciphertext stays valid
        wdk = dk.copy()
        wdk[3::4, rng.integers(0, 384 * k)] ^= 1  # every fourth key has another secret vector
        got = ml.decaps(bad, wdk)
        oek, odk, oct_, oss, odec = (np.empty_like(x) for x in (ek, dk, ct, ss, ss))

        def work(lo, hi):
            oek[lo:hi], odk[lo:hi] = oracle.mlkem_keypair(k, kc[lo:hi])
            oct_[lo:hi], oss[lo:hi] = oracle.mlkem_enc(k, oek[lo:hi], mc[lo:hi])
            odec[lo:hi] = oracle.mlkem_dec(k, bad[lo:hi], wdk[lo:hi])

        b = np.linspace(0, n, min(threads, n) + 1).astype(int)
        ths = [threading.Thread(target=work, args=(b[i], b[i + 1])) for i in range(len(b) - 1)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for name, a, r in (("ek", ek, oek), ("dk", dk, odk), ("ct", ct, oct_), ("ss", ss, oss), ("dec", got, odec)):
            if not np.array_equal(a, r):
                i = int(np.nonzero((a != r).any(axis=1))[0][0])
                print("MISMATCH ML-KEM-%d %s instance %d of round %d, n = %d (seed %d)" % (256 * k, name, i, rounds, n, seed), flush=True)
                sys.exit(1)
        total += n
        rounds += 1
        if rounds % 10 == 0:
            print("ML-KEM-%d: %d instances identical to the oracle (%.0f s, %d rounds)" % (256 * k, total, time.time() - t0, rounds), flush=True)
    print("ML-KEM-%d: done, %d instances in %d rounds identical to the oracle" % (256 * k, total, rounds), flush=True)
print("soak ok")
