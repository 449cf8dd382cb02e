#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (kyber_batch_enc /
kyber_batch_dec on numpy arrays): what a caller with keys and ciphertexts in
host memory sees.  Not the headline number (bench.py keeps data in HBM)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import kyber_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
kem = kyber_amd.KyberHip(3)
rng = np.random.default_rng(1)
kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
ec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
pk, sk = kem.keypair(kc)
kem.enc(pk[:1024], ec[:1024])
from kyber_amd import lib as kl  # noqa: E402
L = kl.library()
import ctypes  # noqa: E402


def run(tag, pk, sk, ec, ct, ss, ss2):
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        assert L.kyber_batch_enc(3, 0, n, p(ct), p(ss), p(pk), p(ec)) == 0
        t1 = time.perf_counter()
        assert L.kyber_batch_dec(3, 0, n, p(ss2), p(ct), p(sk)) == 0
        t2 = time.perf_counter()
        assert np.array_equal(ss, ss2)
        best = min(best, t2 - t0)
        print("%s: enc %.1f ms  dec %.1f ms  -> %.2f M enc+dec/s" % (tag, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n / (t2 - t0) / 1e6))
    print("%s: best %.2f M enc+dec/s, n = %d" % (tag, n / best / 1e6, n))


ct, ss, ss2 = np.empty((n, kem.ct_bytes), np.uint8), np.empty((n, 32), np.uint8), np.empty((n, 32), np.uint8)
run("pageable", pk, sk, ec, ct, ss, ss2)
ppk, psk, pec = kl.pinned_empty(pk.shape), kl.pinned_empty(sk.shape), kl.pinned_empty(ec.shape)
ppk[:], psk[:], pec[:] = pk, sk, ec
pct, pss, pss2 = kl.pinned_empty(ct.shape), kl.pinned_empty(ss.shape), kl.pinned_empty(ss2.shape)
run("pinned  ", ppk, psk, pec, pct, pss, pss2)
assert np.array_equal(pct, ct)
