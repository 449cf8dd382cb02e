#!/usr/bin/env python3
"""Device-resident batched keypair rate (side measurement; bench.py's metric is encaps+decaps)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import kyber_amd  # noqa: E402
from kyber_amd import lib as kl  # noqa: E402

n = 1 << 20
dev = torch.device("cuda", 0)
for k in (2, 3, 4):
    kem = kyber_amd.KyberHip(k)
    coins = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device=dev)
    pk, sk = kem.keypair_dev(coins)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        kem.keypair_dev(coins, pk, sk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    kl.profile_enable(True)
    kem.keypair_dev(coins, pk, sk)
    prof = kl.profile_read()
    kl.profile_enable(False)
    print("Kyber%d keypair: %.1f M/s  " % (256 * k, n / dt / 1e6) + " ".join("%s=%.3f" % (a.replace("k_", ""), ms / c) for a, (c, ms) in prof.items()))
