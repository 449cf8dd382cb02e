"""GPU: kyber_hip_set_workspace_slots (include/kyber_batch.h).  With several workspaces per device, small batches
that arrive on different streams from different host threads overlap on the device; every byte must be what the
one-workspace library gives (itself pinned by the oracle, the vectors and the soak), for the SHAKE sets, a 90s
set (whose second stream stays unused in this mode: tools/slots_stress.py), ML-KEM, and with a large batch (always slot 0) in the mix."""
import threading

import numpy as np
import pytest
import torch

import kyber_amd
from kyber_amd import lib as kl

pytestmark = pytest.mark.gpu


def _job(kind, k, n90, n, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    if kind == "ml":
        ml = kyber_amd.MlKemHip(k)

        def run():
            ek, dk = ml.keypair_dev(kc)
            ct, ss = ml.encaps_dev(ek, ec)
            bad = ct.clone()
            bad[::3, 5] ^= 1
            return ek, dk, ct, ss, ml.decaps_dev(bad, dk)
    else:
        kem = kyber_amd.KyberHip(k, n90)

        def run():
            pk, sk = kem.keypair_dev(kc)
            ct, ss = kem.enc_dev(pk, ec)
            bad = ct.clone()
            bad[::3, 5] ^= 1
            return pk, sk, ct, ss, kem.dec_dev(bad, sk)
    return run


def _join_all(threads, seconds=150):
    """A stream that never drains (what tools/slots_stress.py found while this mode was built) must fail the test, not hang it."""
    import time
    deadline = time.time() + seconds
    for t in threads:
        t.join(max(0.0, deadline - time.time()))
    stuck = [t.name for t in threads if t.is_alive()]
    if stuck:
        pytest.fail("threads %r still waiting for their streams after %d s" % (stuck, seconds), pytrace=False)


def test_slots_argument_range():
    L = kl.library()
    assert L.kyber_hip_set_workspace_slots(0) == -1 and L.kyber_hip_set_workspace_slots(9) == -1
    assert L.kyber_hip_set_workspace_slots(8) == 0 and L.kyber_hip_set_workspace_slots(1) == 0


def test_concurrent_streams_on_several_workspaces_give_the_same_bytes(oracle):
    jobs = [("ky", 3, 0, 1, 11), ("ky", 3, 0, 300, 12), ("ky", 2, 0, 5000, 13), ("ky", 4, 0, 20000, 14),
            ("ky", 3, 1, 700, 15), ("ml", 3, 0, 900, 16), ("ky", 3, 0, 70000, 17), ("ky", 3, 0, 40, 18)]
    runs = [_job(*j) for j in jobs]
    want = [tuple(t.clone() for t in r()) for r in runs]   # one workspace, one stream
    torch.cuda.synchronize()
    # the one-workspace results of the first job against the oracle, so that the comparison below is anchored here too
    kc = torch.randint(0, 256, (1, 64), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(11))
    opk, osk = oracle.keypair(3, kc.cpu().numpy())
    assert np.array_equal(want[0][0].cpu().numpy(), opk) and np.array_equal(want[0][1].cpu().numpy(), osk)
    errors = []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            for _ in range(6 if jobs[i][3] >= 20000 else 25):
                with torch.cuda.stream(st):
                    got = runs[i]()
                st.synchronize()
                for a, b in zip(got, want[i]):
                    if not torch.equal(a, b):
                        raise AssertionError("job %r differs from the one-workspace result" % (jobs[i],))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    kl.set_workspace_slots(4)
    try:
        ths = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(len(jobs))]
        for t in ths:
            t.start()
        _join_all(ths)
    finally:
        kl.set_workspace_slots(1)
    assert not errors, errors
    # what the extra workspaces held is released with the rest, and the library comes back
    assert kl.library().kyber_hip_release() == 0
    got = runs[1]()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, want[1]))


def test_host_pointer_calls_from_threads_on_several_contexts(oracle):
    """Host-pointer calls (the reference-shaped API of a multithreaded caller): with several slots each takes a free
    host context (own streams and staging) and they overlap; outputs against the oracle."""
    k = 3
    kem = kyber_amd.KyberHip(k)
    sizes = [1, 1, 7, 200, 1000, 40000]
    data = []
    for i, n in enumerate(sizes):
        rng = np.random.default_rng(500 + i)
        kc, ec = rng.integers(0, 256, (n, 64), dtype=np.uint8), rng.integers(0, 256, (n, 32), dtype=np.uint8)
        pk, sk = oracle.keypair(k, kc)
        ct, ss = oracle.enc(k, pk, ec)
        data.append((kc, ec, pk, sk, ct, ss))
    errors = []

    def worker(i):
        try:
            kc, ec, pk, sk, ct, ss = data[i]
            for _ in range(4 if sizes[i] > 10000 else 20):
                gpk, gsk = kem.keypair(kc)
                gct, gss = kem.enc(gpk, ec)
                gdec = kem.dec(gct, gsk)
                if not (np.array_equal(gpk, pk) and np.array_equal(gsk, sk) and np.array_equal(gct, ct)
                        and np.array_equal(gss, ss) and np.array_equal(gdec, ss)):
                    raise AssertionError("host-pointer job %d (n = %d) differs from the oracle" % (i, sizes[i]))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    kl.set_workspace_slots(4)
    try:
        ths = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(len(sizes))]
        for t in ths:
            t.start()
        _join_all(ths)
    finally:
        kl.set_workspace_slots(1)
    assert not errors, errors
    assert kl.library().kyber_hip_release() == 0
    gpk, gsk = kem.keypair(data[3][0])
    assert np.array_equal(gpk, data[3][2]) and np.array_equal(gsk, data[3][3])


def test_captured_calls_among_several_workspaces(oracle):
    """Stream capture with several workspaces configured: the captured call finds the workspace that an eager call of
    its size laid out -- whichever slot that was -- and replays with the oracle's bytes; eager calls on other streams
    in between (the caller orders them against the replays, here by synchronising)."""
    k, n = 3, 100
    kem = kyber_amd.KyberHip(k)
    rng = np.random.default_rng(99)
    kc, ec = rng.integers(0, 256, (n, 64), dtype=np.uint8), rng.integers(0, 256, (n, 32), dtype=np.uint8)
    opk, osk = oracle.keypair(k, kc)
    oct_, oss = oracle.enc(k, opk, ec)
    pk, sk, dec = (torch.from_numpy(a).cuda() for a in (opk, osk, ec))
    kl.set_workspace_slots(4)
    try:
        busy = torch.cuda.Stream()
        with torch.cuda.stream(busy):          # takes the first slot, with a smaller batch
            kem.enc_dev(pk[:3].contiguous(), dec[:3].contiguous())
        warm = torch.cuda.Stream()
        with torch.cuda.stream(warm):          # lays out another slot for n
            kem.dec_dev(kem.enc_dev(pk, dec)[0], sk)
        torch.cuda.synchronize()
        ct, ss, ssd = (torch.zeros((n, kem.ct_bytes), dtype=torch.uint8, device="cuda"), torch.zeros((n, 32), dtype=torch.uint8, device="cuda"),
                       torch.zeros((n, 32), dtype=torch.uint8, device="cuda"))
        cap, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(graph, stream=cap):
                kem.enc_dev(pk, dec, ct, ss)
                kem.dec_dev(ct, sk, ssd)
        torch.cuda.synchronize()
        assert not bool(ss.any())
        for _ in range(3):
            ct.zero_(), ss.zero_(), ssd.zero_()
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            assert np.array_equal(ct.cpu().numpy(), oct_) and np.array_equal(ss.cpu().numpy(), oss) and np.array_equal(ssd.cpu().numpy(), oss)
            with torch.cuda.stream(busy):
                c2, s2 = kem.enc_dev(pk, dec)
            torch.cuda.synchronize()
            assert np.array_equal(c2.cpu().numpy(), oct_) and np.array_equal(s2.cpu().numpy(), oss)
    finally:
        kl.set_workspace_slots(1)
