#!/usr/bin/env python3
"""Soak test of the multi-workspace mode (kyber_hip_set_workspace_slots): several host threads call the
host-pointer entry points at once, each with batches of random size, and every byte of every call is compared
with the reference's own compiled code (oracle/_ref, avx2 or ref) run by the same thread.
Test infrastructure (uses oracle_lib.py beside it), run by hand, not collected by pytest.

    python3 tests/soak_threads.py [seconds per set = 40] [threads = 8] [largest batch = 20000] [seed shift = 0]
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
from kyber_amd import lib as kl  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
shift = int(sys.argv[4]) if len(sys.argv) > 4 else 0
with open(kyber_amd.lib.library_path(), "rb") as f:
    print("library sha256 %s, %d threads, %d workspaces, seed shift %d" % (hashlib.sha256(f.read()).hexdigest()[:12], threads, min(threads, 8), shift), flush=True)
kl.set_workspace_slots(min(threads, 8))
impl = "avx2" if oracle_lib.has_avx2() else "ref"
for k in (3, 2, 4):
    for n90 in (0, 1):
        name = "Kyber%d%s" % (256 * k, "-90s" if n90 else "")
        kem = kyber_amd.KyberHip(k, n90)
        totals, calls, errors = [0] * threads, [0] * threads, []
        t0 = time.time()

        def work(t):
            ref = oracle_lib.RefLib(impl, k, n90)
            rng = np.random.default_rng(3_000_017 * k + n90 + 101 * t + 7919 * shift)
            while time.time() - t0 < budget and not errors:
                n = int(rng.integers(1, nmax + 1)) if rng.random() < 0.7 else int(rng.integers(1, 33))  # many tiny batches too
                kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
                ec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
                pk, sk = kem.keypair(kc)
                ct, ss = kem.enc(pk, ec)
                bad = ct.copy()
                bad[::2, int(rng.integers(0, ct.shape[1]))] ^= 0x10
                got = kem.dec(bad, sk)
                rpk, rsk = ref.keypair(kc)
                rct, rss = ref.enc(rpk, ec)
                rdec = ref.dec(bad, rsk)
                for what, a, r in (("pk", pk, rpk), ("sk", sk, rsk), ("ct", ct, rct), ("ss", ss, rss), ("dec", got, rdec)):
                    if not np.array_equal(a, r):
                        errors.append("MISMATCH %s %s, thread %d, call %d, n = %d" % (name, what, t, calls[t], n))
                        return
                totals[t] += n
                calls[t] += 1

        ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        if errors:
            print(errors[0], flush=True)
            kl.set_workspace_slots(1)
            sys.exit(1)
        print("%s: done, %d instances in %d calls from %d threads identical to %s (%.0f s)" % (name, sum(totals), sum(calls), threads, impl, time.time() - t0), flush=True)
kl.set_workspace_slots(1)
print("soak ok")
