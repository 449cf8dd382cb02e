#!/usr/bin/env python3
"""Small batches from several host threads, each on its own stream: calls per second with one workspace per
device (the default: the GPU work of the calls is ordered) and with kyber_hip_set_workspace_slots(threads)."""
import ctypes
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kyber_amd  # noqa: E402
from kyber_amd import lib as kl  # noqa: E402

L = kl.library()
K = 3
kem = kyber_amd.KyberHip(K)
d = lambda t: ctypes.c_void_p(t.data_ptr())
CALLS = 300


def measure(threads, n, op):
    ctx = []
    for t in range(threads):
        g = torch.Generator(device="cuda")
        g.manual_seed(1000 * t + n)
        kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
        ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
        pk, sk = kem.keypair_dev(kc)
        ct, ss = kem.enc_dev(pk, ec)
        ctx.append((pk, sk, ct, ss, ec, torch.empty_like(ss), torch.empty_like(ct), torch.cuda.Stream()))
    torch.cuda.synchronize()
    bar = threading.Barrier(threads + 1)

    def worker(t):
        pk, sk, ct, ss, ec, ss2, ct2, st = ctx[t]
        sp = ctypes.c_void_p(st.cuda_stream)
        bar.wait()
        for _ in range(CALLS):
            if op == "dec":
                assert L.kyber_batch_dec_dev(K, 0, n, d(ss2), d(ct), d(sk), sp) == 0
            else:
                assert L.kyber_batch_enc_dev(K, 0, n, d(ct2), d(ss2), d(pk), d(ec), sp) == 0
            st.synchronize()
        assert torch.equal(ss2, ss)
        bar.wait()

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for t in ths:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in ths:
        t.join()
    return threads * CALLS / dt


if len(sys.argv) > 1:  # tools/concurrency_rate.py <threads> <n> <enc|dec> <slots>: one measurement (e.g. under rocprofv3 --kernel-trace)
    T, N, OP, S = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    kl.set_workspace_slots(S)
    print("%s n=%d, %d threads, %d workspaces: %.0f calls/s" % (OP, N, T, S, measure(T, N, OP)), flush=True)
    kl.set_workspace_slots(1)
    sys.exit(0)

import numpy as np  # noqa: E402

hp = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def measure_host(threads, n, op):
    """the same with host pointers (pageable memory): what a multithreaded caller of the reference-shaped API does"""
    ctx = []
    for t in range(threads):
        rng = np.random.default_rng(1000 * t + n)
        kc, ec = rng.integers(0, 256, (n, 64), dtype=np.uint8), rng.integers(0, 256, (n, 32), dtype=np.uint8)
        pk, sk = kem.keypair(kc)
        ct, ss = kem.enc(pk, ec)
        ctx.append((pk, sk, ct, ss, ec, np.empty_like(ss), np.empty_like(ct)))
    bar = threading.Barrier(threads + 1)

    def worker(t):
        pk, sk, ct, ss, ec, ss2, ct2 = ctx[t]
        bar.wait()
        for _ in range(CALLS):
            if op == "dec":
                assert L.kyber_batch_dec(K, 0, n, hp(ss2), hp(ct), hp(sk)) == 0
            else:
                assert L.kyber_batch_enc(K, 0, n, hp(ct2), hp(ss2), hp(pk), hp(ec)) == 0
        assert np.array_equal(ss2, ss)
        bar.wait()

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for t in ths:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in ths:
        t.join()
    return threads * CALLS / dt


for op in ("enc", "dec"):
    for n in (1, 256):
        row = []
        for threads in (1, 2, 4, 8):
            kl.set_workspace_slots(1)
            a = measure_host(threads, n, op)
            kl.set_workspace_slots(min(threads, 8))
            b = measure_host(threads, n, op)
            kl.set_workspace_slots(1)
            row.append("%d thr %6.0f -> %6.0f" % (threads, a, b))
        print("%s n=%5d  host pointers, calls/s, 1 workspace -> one per thread:  " % (op, n) + " | ".join(row), flush=True)

for op in ("enc", "dec"):
    for n in (1, 256, 4096, 16384):
        row = []
        for threads in (1, 2, 4, 8):
            kl.set_workspace_slots(1)
            a = measure(threads, n, op)
            kl.set_workspace_slots(min(threads, 8))
            b = measure(threads, n, op)
            kl.set_workspace_slots(1)
            row.append("%d thr %6.0f -> %6.0f" % (threads, a, b))
        print("%s n=%5d  calls/s, 1 workspace -> one per thread:  " % (op, n) + " | ".join(row), flush=True)
