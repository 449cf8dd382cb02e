"""Test-only stand-in for the HIP engine so that bench.py's N-rank plumbing
(launcher, rendezvous, barrier, max-over-ranks, one JSON line) can run on CPU:
the same *_dev method names over CPU torch tensors, computed by the oracle.
Selected with KYBER_BENCH_ENGINE=bench_stub_engine:make and --device cpu."""
import numpy as np
import torch

import oracle_lib


class StubEngine:
    def __init__(self, k, nineties):
        self.k, self.nineties = k, int(nineties)
        self.o = oracle_lib.load_oracle()
        self.pk_bytes, self.sk_bytes, self.ct_bytes = self.o.sizes(k)

    def keypair_dev(self, coins):
        pk, sk = self.o.keypair(self.k, coins.numpy(), self.nineties)
        return torch.from_numpy(pk), torch.from_numpy(sk)

    def enc_dev(self, pk, coins, ct, ss):
        c, s = self.o.enc(self.k, pk.numpy(), coins.numpy(), self.nineties)
        ct.copy_(torch.from_numpy(c))
        ss.copy_(torch.from_numpy(s))
        return ct, ss

    def dec_dev(self, ct, sk, ss):
        ss.copy_(torch.from_numpy(self.o.dec(self.k, ct.numpy(), sk.numpy(), self.nineties)))
        return ss


def make(k, nineties):
    return StubEngine(k, nineties)


def make_failing_rank1(k, nineties):
    """rank 1 dies while rank 0 carries on into the barrier: what the launcher must not wait out"""
    import os
    if os.environ.get("RANK") == "1":
        raise SystemExit(7)
    return StubEngine(k, nineties)
