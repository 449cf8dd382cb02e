"""GPU: kyber_batch_dec_dev_noalloc (the entry point that allocates nothing,
include/kyber_batch.h) gives the same bytes as kyber_batch_dec_dev, on two
calls in a row that share the library's fixed scratch array -- the second with
a smaller batch -- for every parameter set; oversize batches are refused."""
import ctypes

import numpy as np
import pytest
import torch

import kyber_amd
from kyber_amd import lib as kl

pytestmark = pytest.mark.gpu

SETS = [(2, 0), (3, 0), (4, 0), (2, 1), (3, 1), (4, 1)]


def _instances(kem, n, seed, oracle):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    pk, sk = kem.keypair_dev(kc)
    ct, ss = kem.enc_dev(pk, ec)
    ct[::5, 3] ^= 0x40  # every fifth ciphertext is rejected (implicit rejection path, ref/kem.c:116-122)
    return ct, sk


@pytest.mark.parametrize("k,n90", SETS)
def test_noalloc_matches_dec_dev_two_calls(k, n90, oracle):
    kem = kyber_amd.KyberHip(k, n90)
    nmax = kl.noalloc_max_n()
    for call, n in enumerate((nmax, 777)):  # full scratch first, then a smaller batch over its leftovers
        ct, sk = _instances(kem, n, 100 * k + n90 + call, oracle)
        want = kem.dec_dev(ct, sk)
        got = torch.full((n, 32), 0xA5, dtype=torch.uint8, device="cuda")
        kem.dec_dev_noalloc(ct, sk, got)
        torch.cuda.synchronize()
        assert torch.equal(got, want), "call %d (n=%d) differs from kyber_batch_dec_dev" % (call, n)
        idx = torch.arange(0, n, 97, device="cuda")
        assert np.array_equal(got[idx].cpu().numpy(), oracle.dec(k, ct[idx].cpu().numpy(), sk[idx].cpu().numpy(), n90))


def test_noalloc_rejects_oversize_and_bad_arguments():
    kem = kyber_amd.KyberHip(3)
    n = kl.noalloc_max_n() + 1
    ct = torch.zeros((n, kem.ct_bytes), dtype=torch.uint8, device="cuda")
    sk = torch.zeros((n, kem.sk_bytes), dtype=torch.uint8, device="cuda")
    ss = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    with pytest.raises(kl.KyberHipError, match="KYBER_HIP_NOALLOC_MAX_N"):
        kem.dec_dev_noalloc(ct, sk, ss)
    L = kl.library()
    assert L.kyber_batch_dec_dev_noalloc(3, 0, 1, None, ctypes.c_void_p(ct.data_ptr()), ctypes.c_void_p(sk.data_ptr()), None) == -1
    assert L.kyber_batch_dec_dev_noalloc(5, 0, 1, ctypes.c_void_p(ss.data_ptr()), ctypes.c_void_p(ct.data_ptr()),
                                         ctypes.c_void_p(sk.data_ptr()), None) == -1
    assert L.kyber_batch_dec_dev_noalloc(3, 0, 0, ctypes.c_void_p(ss.data_ptr()), ctypes.c_void_p(ct.data_ptr()),
                                         ctypes.c_void_p(sk.data_ptr()), None) == 0


@pytest.mark.parametrize("n90", [0, 1])
@pytest.mark.parametrize("n", [1, 9, 1000, 1024])
def test_noalloc_can_be_captured_in_a_graph(oracle, n90, n):
    """The callers this entry point exists for: under stream capture nothing may allocate.  The
    first call (outside capture) takes the scratch array's address; a captured call replays
    with the same bytes as kyber_batch_dec_dev -- on EVERY replay (round 3: a memset node in the
    captured graph worked on the first launch only and corrupted instance 0 from the second on
    whenever n was not a multiple of 8; the graph now holds kernels only) -- also after an input
    changed in place.  Both the SHAKE and the 90s kernel sequences (the AES launches pick their
    workgroup size on the host)."""
    kem = kyber_amd.KyberHip(3, n90)
    ct, sk = _instances(kem, n, 4242, oracle)
    want = kem.dec_dev(ct, sk)
    out = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    kem.dec_dev_noalloc(ct, sk, out)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    out.zero_()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            kem.dec_dev_noalloc(ct, sk, out)
    torch.cuda.synchronize()
    assert int(out.sum()) == 0            # capture records, it does not run
    for _ in range(4):
        out.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
    kem.dec_dev_noalloc(ct, sk, out)      # an eager call between replays
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    ct[0, 7] ^= 0x10                      # an input changes in place: instance 0's (implicit-rejection) secret changes with H(ct)
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, kem.dec_dev(ct, sk)) and not torch.equal(out[0], want[0])


def test_noalloc_eager_calls_on_two_streams_are_ordered(oracle):
    """Outside capture the library orders calls that share the scratch array: back-to-back calls on
    two streams, with nothing between them, both give kyber_batch_dec_dev's bytes."""
    kem = kyber_amd.KyberHip(3)
    n = kl.noalloc_max_n()
    cta, ska = _instances(kem, n, 91, oracle)
    ctb, skb = _instances(kem, n, 92, oracle)
    wa, wb = kem.dec_dev(cta, ska), kem.dec_dev(ctb, skb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        oa = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
        ob = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            kem.dec_dev_noalloc(cta, ska, oa)
        with torch.cuda.stream(s2):
            kem.dec_dev_noalloc(ctb, skb, ob)
        with torch.cuda.stream(s1):
            kem.dec_dev_noalloc(ctb, skb, ob)
        torch.cuda.synchronize()
        assert torch.equal(oa, wa) and torch.equal(ob, wb)


def test_release_wipes_and_everything_comes_back(oracle):
    """kyber_hip_release covers every device the library has touched and the scratch array's
    contents; the next calls rebuild what they need"""
    kem = kyber_amd.KyberHip(3, 1)
    ct, sk = _instances(kem, 500, 7, oracle)
    want = kem.dec_dev(ct, sk)
    out = torch.zeros((500, 32), dtype=torch.uint8, device="cuda")
    kem.dec_dev_noalloc(ct, sk, out)
    assert kl.library().kyber_hip_release() == 0
    assert kl.library().kyber_hip_release() == 0
    out.zero_()
    kem.dec_dev_noalloc(ct, sk, out)
    torch.cuda.synchronize()
    assert torch.equal(out, want) and torch.equal(kem.dec_dev(ct, sk), want)
