#!/usr/bin/env python3
"""Latency of small host-pointer batches (the reference-shaped batch-of-one API
and small n): microseconds per call, median of 200."""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import kyber_amd  # noqa: E402
from kyber_amd import lib as kl  # noqa: E402

L = kl.library()
N90 = int(len(sys.argv) > 1 and sys.argv[1] == "90s")   # tools/latency.py 90s: Kyber768-90s
kem = kyber_amd.KyberHip(3, N90)
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
SIZES = tuple(int(x) for x in __import__("os").environ.get("KY_LAT_SIZES", "1,16,256,4096,8192,16384").split(","))  # KY_LAT_SIZES=32768,65536: other sizes
for n in SIZES:
    rng = np.random.default_rng(n)
    kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = kem.keypair(kc)
    ct, ss = kem.enc(pk, ec)
    ss2 = np.empty_like(ss)
    res = {}
    for name, fn in (("keypair", lambda: L.kyber_batch_keypair(3, N90, n, p(pk), p(sk), p(kc))),
                     ("enc", lambda: L.kyber_batch_enc(3, N90, n, p(ct), p(ss), p(pk), p(ec))),
                     ("dec", lambda: L.kyber_batch_dec(3, N90, n, p(ss2), p(ct), p(sk)))):
        ts = []
        for _ in range(200):
            t0 = time.perf_counter()
            assert fn() == 0
            ts.append(time.perf_counter() - t0)
        res[name] = np.median(ts) * 1e6
    assert np.array_equal(ss, ss2)
    print("n=%5d  keypair %7.1f us  enc %7.1f us  dec %7.1f us" % (n, res["keypair"], res["enc"], res["dec"]))

# one key, many operations: the shared-key calls against the same calls through a prepared handle
# (kyber_hip_prepare_pk / _sk: expanded matrix and H(pk) kept on the device)
pk1, sk1 = kem.keypair(np.random.default_rng(1).integers(0, 256, (1, 64), dtype=np.uint8))
with kem.prepare_pk(pk1[0]) as ppk, kem.prepare_sk(sk1[0]) as psk:
    for n in (1, 256, 4096):
        ec = np.random.default_rng(n).integers(0, 256, (n, 32), dtype=np.uint8)
        ct, ss = kem.enc_shared_pk(pk1[0], ec)
        ct2, ss2, ss3 = np.empty_like(ct), np.empty_like(ss), np.empty_like(ss)
        res = {}
        for name, fn in (("enc_shared", lambda: L.kyber_batch_enc_shared_pk(3, N90, n, p(ct2), p(ss2), p(pk1), p(ec))),
                         ("enc_prepared", lambda: L.kyber_batch_enc_prepared(ppk._h, n, p(ct2), p(ss2), p(ec))),
                         ("dec_shared", lambda: L.kyber_batch_dec_shared_sk(3, N90, n, p(ss3), p(ct), p(sk1))),
                         ("dec_prepared", lambda: L.kyber_batch_dec_prepared(psk._h, n, p(ss3), p(ct)))):
            ts = []
            for _ in range(200):
                t0 = time.perf_counter()
                assert fn() == 0
                ts.append(time.perf_counter() - t0)
            res[name] = np.median(ts) * 1e6
            assert np.array_equal(ct, ct2) and np.array_equal(ss, ss2)
        assert np.array_equal(ss, ss3)
        print("n=%5d  one key: enc shared %7.1f us  prepared %7.1f us | dec shared %7.1f us  prepared %7.1f us"
              % (n, res["enc_shared"], res["enc_prepared"], res["dec_shared"], res["dec_prepared"]))

# data resident in HBM: the device-pointer calls, enqueue + synchronise (what a serving loop that
# keeps keys and ciphertexts on the device pays per batch)
import torch  # noqa: E402

for n in (1, 256, 4096, 16384, 65536):
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    pk, sk = kem.keypair_dev(kc)
    ct, ss = kem.enc_dev(pk, ec)
    ss2 = torch.empty_like(ss)
    res = {}
    for name, fn in (("keypair", lambda: kem.keypair_dev(kc, pk, sk)), ("enc", lambda: kem.enc_dev(pk, ec, ct, ss)),
                     ("dec", lambda: kem.dec_dev(ct, sk, ss2))):
        ts = []
        st = torch.cuda.current_stream()
        for _ in range(200):
            st.synchronize()
            t0 = time.perf_counter()
            fn()
            st.synchronize()  # the caller's stream, not the device: what a serving loop waits for
            ts.append(time.perf_counter() - t0)
        res[name] = np.median(ts) * 1e6
    assert torch.equal(ss, ss2)
    print("n=%5d  device pointers: keypair %7.1f us  enc %7.1f us  dec %7.1f us" % (n, res["keypair"], res["enc"], res["dec"]))
