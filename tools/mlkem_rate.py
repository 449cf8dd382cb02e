#!/usr/bin/env python3
"""Device-resident batched ML-KEM rates (side measurement; bench.py's metric is round-3 Kyber768 encaps+decaps):
2^20 instances per step, keygen and encaps+decaps per second for the three parameter sets, with the
per-kernel times of one step and the round-3 set measured beside it on the same box."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kyber_amd  # noqa: E402
from kyber_amd import lib as kl  # noqa: E402

L = kl.library()
n = 1 << 20
dev = torch.device("cuda", 0)
d = lambda t: ctypes.c_void_p(t.data_ptr())
g = torch.Generator(device=dev)
g.manual_seed(203)
kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device=dev, generator=g)
mc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def rate(fn, steps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return n * steps / (time.perf_counter() - t0)


def kernels(fn):
    kl.profile_enable(True)
    fn()
    prof = kl.profile_read()
    kl.profile_enable(False)
    return " ".join("%s=%.3f" % (a.replace("k_", ""), ms / c) for a, (c, ms) in prof.items())


for k in (2, 3, 4):
    ml, kem = kyber_amd.MlKemHip(k), kyber_amd.KyberHip(k)
    ek, dk = ml.keypair_dev(kc)
    ct, ss = ml.encaps_dev(ek, mc)
    ss2 = torch.empty_like(ss)

    def ml_kp():
        assert L.mlkem_batch_keypair_dev(k, n, d(ek), d(dk), d(kc), sp) == 0

    def ml_encdec():
        assert L.mlkem_batch_encaps_dev(k, n, d(ct), d(ss), d(ek), d(mc), sp) == 0
        assert L.mlkem_batch_decaps_dev(k, n, d(ss2), d(ct), d(dk), sp) == 0

    pk, sk = kem.keypair_dev(kc)
    kct, kss = kem.enc_dev(pk, mc)
    kss2 = torch.empty_like(kss)

    def ky_encdec():
        kem.enc_dev(pk, mc, kct, kss)
        kem.dec_dev(kct, sk, kss2)

    r_kp, r_ed, r_ky = rate(ml_kp), rate(ml_encdec), rate(ky_encdec)
    assert torch.equal(ss, ss2) and torch.equal(kss, kss2)
    print("ML-KEM-%d: keygen %.1f M/s, encaps+decaps %.1f M/s (round-3 Kyber%d on this box: %.1f M/s)" % (256 * k, r_kp / 1e6, r_ed / 1e6, 256 * k, r_ky / 1e6), flush=True)
    print("   one encaps+decaps step, ms per launch of 2^19: " + kernels(ml_encdec), flush=True)
