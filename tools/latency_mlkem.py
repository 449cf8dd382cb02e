#!/usr/bin/env python3
"""Latency of small ML-KEM batches (include/mlkem_batch.h), tools/latency_mlkem.py [k = 3]: ML-KEM-512/768/1024
for k = 2/3/4; microseconds per call, median of 200,
host pointers and device pointers (enqueue + synchronise of the caller's stream).
KY_LAT_SIZES=1,256,...: other sizes."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kyber_amd  # noqa: E402
from kyber_amd import lib as kl  # noqa: E402

L = kl.library()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
print("ML-KEM-%d" % (256 * K), flush=True)
ml = kyber_amd.MlKemHip(K)
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
d = lambda t: ctypes.c_void_p(t.data_ptr())
SIZES = tuple(int(x) for x in os.environ.get("KY_LAT_SIZES", "1,16,256,4096,16384,65536").split(","))


def median_us(fn, sync=None):
    ts = []
    for _ in range(200):
        if sync:
            sync()
        t0 = time.perf_counter()
        assert fn() == 0
        if sync:
            sync()
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6


for n in SIZES:
    rng = np.random.default_rng(n)
    kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    mc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ek, dk = ml.keypair(kc)
    ct, ss = ml.encaps(ek, mc)
    ss2 = np.empty_like(ss)
    r = (median_us(lambda: L.mlkem_batch_keypair(K, n, p(ek), p(dk), p(kc))),
         median_us(lambda: L.mlkem_batch_encaps(K, n, p(ct), p(ss), p(ek), p(mc))),
         median_us(lambda: L.mlkem_batch_decaps(K, n, p(ss2), p(ct), p(dk))))
    assert np.array_equal(ss, ss2)
    print("n=%5d  host pointers:   keygen %7.1f us  encaps %7.1f us  decaps %7.1f us" % ((n,) + r), flush=True)
    tkc, tmc = torch.from_numpy(kc).cuda(), torch.from_numpy(mc).cuda()
    tek, tdk = ml.keypair_dev(tkc)
    tct, tss = ml.encaps_dev(tek, tmc)
    tss2 = torch.empty_like(tss)
    st = torch.cuda.current_stream()
    sp = ctypes.c_void_p(st.cuda_stream)
    r = (median_us(lambda: L.mlkem_batch_keypair_dev(K, n, d(tek), d(tdk), d(tkc), sp), st.synchronize),
         median_us(lambda: L.mlkem_batch_encaps_dev(K, n, d(tct), d(tss), d(tek), d(tmc), sp), st.synchronize),
         median_us(lambda: L.mlkem_batch_decaps_dev(K, n, d(tss2), d(tct), d(tdk), sp), st.synchronize))
    assert torch.equal(tss, tss2) and np.array_equal(tss.cpu().numpy(), ss) and np.array_equal(tct.cpu().numpy(), ct)
    print("n=%5d  device pointers: keygen %7.1f us  encaps %7.1f us  decaps %7.1f us" % ((n,) + r), flush=True)
