"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- rendezvous, disjoint
contiguous shards, no data-path collective, max-over-ranks timing.  The compute
callback here is the oracle (test-only stand-in for the GPU engine)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from kyber_amd import dist as kdist
    import oracle_lib
    kdist.init("gloo")
    o = oracle_lib.load_oracle()
    rng = np.random.default_rng(42)  # same coins on every rank; each takes its slice
    kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ec = rng.integers(0, 256, (n, 32), dtype=np.uint8)

    def compute(lo, hi):
        pk, sk = o.keypair(3, kc[lo:hi])
        ct, ss = o.enc(3, pk, ec[lo:hi])
        assert np.array_equal(o.dec(3, ct, sk), ss)
        return ct, ss

    dist.barrier()
    lo, hi, (ct, ss) = kdist.run_sharded(n, compute)
    dist.barrier()
    slow = kdist.max_over_ranks(1.0 + rank)
    assert slow == float(world)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), lo=lo, hi=hi, ct=ct, ss=ss)
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path, oracle):
    n, world = 37, 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(42)
    kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = oracle.keypair(3, kc)
    ct, ss = oracle.enc(3, pk, ec)
    parts = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    assert parts[0]["lo"] == 0 and parts[-1]["hi"] == n and parts[0]["hi"] == parts[1]["lo"]
    assert np.array_equal(np.concatenate([p["ct"] for p in parts]), ct)
    assert np.array_equal(np.concatenate([p["ss"] for p in parts]), ss)
