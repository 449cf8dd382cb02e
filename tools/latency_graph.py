#!/usr/bin/env python3
"""Decapsulation through kyber_batch_dec_dev_noalloc: eager calls against replays of a captured HIP graph
(torch.cuda.CUDAGraph), microseconds per batch, median of 200, Kyber768, data resident in HBM."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import kyber_amd  # noqa: E402

kem = kyber_amd.KyberHip(3, int(len(sys.argv) > 1 and sys.argv[1] == "90s"))
for n in (1, 256, 4096):
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    pk, sk = kem.keypair_dev(kc)
    ct, ss = kem.enc_dev(pk, ec)
    out = torch.zeros_like(ss)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        kem.dec_dev_noalloc(ct, sk, out)  # first call outside capture: takes the scratch array's address
    side.synchronize()
    assert torch.equal(out, ss)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            kem.dec_dev_noalloc(ct, sk, out)
    res = {}
    for name, fn, sync in (("eager dec_dev", lambda: kem.dec_dev(ct, sk, out), torch.cuda.current_stream().synchronize),
                           ("eager noalloc", lambda: kem.dec_dev_noalloc(ct, sk, out), torch.cuda.current_stream().synchronize),
                           ("graph replay", graph.replay, torch.cuda.synchronize)):
        ts = []
        for _ in range(200):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            sync()
            ts.append(time.perf_counter() - t0)
        res[name] = np.median(ts) * 1e6
        assert torch.equal(out, ss), (n, name, int((out != ss).any(dim=1).sum()))
    print("n=%5d  dec: " % n + "  ".join("%s %6.1f us" % kv for kv in res.items()))
