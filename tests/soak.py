#!/usr/bin/env python3
"""Soak test: GPU against the reference's own compiled code (oracle/_ref, avx2 or
ref) on millions of random instances per parameter set, every byte compared.
Test infrastructure (uses oracle_lib.py beside it), run by hand, not collected by pytest.

    python3 tests/soak.py [seconds per set = 60] [batch = 2^19]
    python3 tests/soak.py 40 -40000      # a negative batch: every round draws its size from [1, 40000] (the small-batch routes)
    python3 tests/soak.py 90 524288 7    # a third argument moves every set's seed: other inputs than the last run's
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
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 19
shift = int(sys.argv[3]) if len(sys.argv) > 3 else 0
threads = 16
with open(kyber_amd.lib.library_path(), "rb") as f:
    print("library sha256 %s, seed shift %d" % (hashlib.sha256(f.read()).hexdigest()[:12], shift), flush=True)
for k in (3, 2, 4):
    for n90 in (0, 1):
        impl = "avx2" if oracle_lib.has_avx2() else "ref"
        ref = oracle_lib.RefLib(impl, k, n90)
        kem = kyber_amd.KyberHip(k, n90)
        seed = 1_000_003 * k + n90 + 7919 * shift
        rng = np.random.default_rng(seed)
        t0, total, rounds = time.time(), 0, 0
        while time.time() - t0 < budget:
            n = nmax if nmax > 0 else int(rng.integers(1, -nmax + 1))
            kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
            ec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            pk, sk = kem.keypair(kc)
            ct, ss = kem.enc(pk, ec)
            bad = ct.copy()
            pos = rng.integers(0, ct.shape[1], size=n)
            bad[np.arange(n), pos] ^= rng.integers(1, 256, size=n, dtype=np.uint8)
            bad[::2] = ct[::2]
            got = kem.dec(bad, sk)
            rpk, rsk, rct, rss, rdec = (np.empty_like(x) for x in (pk, sk, ct, ss, ss))

            def work(lo, hi):
                rpk[lo:hi], rsk[lo:hi] = ref.keypair(kc[lo:hi])
                rct[lo:hi], rss[lo:hi] = ref.enc(rpk[lo:hi], ec[lo:hi])
                rdec[lo:hi] = ref.dec(bad[lo:hi], rsk[lo:hi])

            b = np.linspace(0, n, min(threads, n) + 1).astype(int)
            ths = [threading.Thread(target=work, args=(b[i], b[i + 1])) for i in range(len(b) - 1)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            for name, a, r in (("pk", pk, rpk), ("sk", sk, rsk), ("ct", ct, rct), ("ss", ss, rss), ("dec", got, rdec)):
                if not np.array_equal(a, r):
                    i = int(np.nonzero((a != r).any(axis=1))[0][0])
                    print("MISMATCH Kyber%d%s %s instance %d of round %d (seed %d)" % (256 * k, "-90s" if n90 else "", name, i, rounds, seed), flush=True)
                    sys.exit(1)
            total += n
            rounds += 1
            if nmax > 0 or rounds % 50 == 0:
                print("Kyber%d%s: %d instances identical to %s (%.0f s, %d rounds)" % (256 * k, "-90s" if n90 else "", total, impl, time.time() - t0, rounds), flush=True)
        print("Kyber%d%s: done, %d instances in %d rounds identical to %s" % (256 * k, "-90s" if n90 else "", total, rounds, impl), flush=True)
print("soak ok")
