"""GPU parity: libkyber_hip.so (through the C-ABI) against the oracle on the
same coins -- bit-exact pk, sk, ct, ss for every SHAKE parameter set, ragged
batch sizes, and the reference's negative tests (ref/test_kyber.c:34-95)."""
import numpy as np
import pytest

import kyber_amd

pytestmark = pytest.mark.gpu


def _coins(n, width, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(n, width), dtype=np.uint8)


@pytest.mark.parametrize("n90", [0, 1])
@pytest.mark.parametrize("k", [2, 3, 4])
@pytest.mark.parametrize("n", [1, 7, 8, 9, 64, 257, 1000])
def test_kem_matches_oracle(oracle, k, n, n90):
    kem = kyber_amd.KyberHip(k, n90)
    kc, ec = _coins(n, 64, 100 * k + n), _coins(n, 32, 200 * k + n)
    pk, sk = kem.keypair(kc)
    opk, osk = oracle.keypair(k, kc, n90)
    assert np.array_equal(pk, opk), "pk differs"
    assert np.array_equal(sk, osk), "sk differs"
    ct, ss = kem.enc(pk, ec)
    oct_, oss = oracle.enc(k, opk, ec, n90)
    assert np.array_equal(ct, oct_), "ct differs"
    assert np.array_equal(ss, oss), "ss (enc) differs"
    ss2 = kem.dec(ct, sk)
    assert np.array_equal(ss2, oss), "ss (dec) differs"


@pytest.mark.parametrize("n90", [0, 1])
@pytest.mark.parametrize("k", [2, 3, 4])
def test_invalid_ciphertext_and_key(oracle, k, n90):
    """ref/test_kyber.c:34-95: a flipped ciphertext byte or a wrong secret key
    must give the implicit-rejection secret, the same one the oracle derives."""
    n = 96
    kem = kyber_amd.KyberHip(k, n90)
    kc, ec = _coins(n, 64, 31 + k), _coins(n, 32, 37 + k)
    pk, sk = kem.keypair(kc)
    ct, ss = kem.enc(pk, ec)
    rng = np.random.default_rng(5 + k)
    bad = ct.copy()
    pos = rng.integers(0, ct.shape[1], size=n)
    bad[np.arange(n), pos] ^= rng.integers(1, 256, size=n, dtype=np.uint8)
    bad[::3] = ct[::3]  # keep a third valid so both branches share wavefronts
    got = kem.dec(bad, sk)
    want = oracle.dec(k, bad, sk, n90)
    assert np.array_equal(got, want)
    assert np.array_equal(got[::3], ss[::3])
    assert not np.any(np.all(got[1::3] == ss[1::3], axis=1))
    wrong_sk = _coins(n, sk.shape[1], 41 + k)
    assert np.array_equal(kem.dec(ct, wrong_sk), oracle.dec(k, ct, wrong_sk, n90))


@pytest.mark.parametrize("cap", [0, 5, None])
def test_matrix_tail_queue_capacity(oracle, cap):
    """The matrix sampler parks entries that need a 4th SHAKE128 block (about
    1 % of them: ~80 of the 9000 here) in a queue for a second kernel; entries
    that do not fit the queue are finished in place.  Every capacity, none
    included, must give the reference's bytes (ref/indcpa.c:164-192)."""
    from kyber_amd import lib as kl
    k, n = 3, 1000
    kem = kyber_amd.KyberHip(k)
    kc, ec = _coins(n, 64, 77), _coins(n, 32, 78)
    try:
        kl.set_tail_capacity(cap)
        pk, sk = kem.keypair(kc)
        ct, ss = kem.enc(pk, ec)
        ss2 = kem.dec(ct, sk)
    finally:
        kl.set_tail_capacity(None)
    opk, osk = oracle.keypair(k, kc)
    oct_, oss = oracle.enc(k, opk, ec)
    assert np.array_equal(pk, opk) and np.array_equal(sk, osk)
    assert np.array_equal(ct, oct_) and np.array_equal(ss, oss) and np.array_equal(ss2, oss)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_multi_device_entry_points_shard_contiguously(oracle, devices):
    """kyber_batch_*_mgpu: contiguous shards, one host thread per listed device.
    On a one-GPU box every shard names device 0 (they queue behind each other);
    the bytes must be those of the single-device call and of the oracle, also
    when n does not divide by the shard count."""
    k, n = 3, 1001
    kem = kyber_amd.KyberHip(k)
    kc, ec = _coins(n, 64, 501), _coins(n, 32, 502)
    pk, sk = kem.keypair_mgpu(kc, devices)
    ct, ss = kem.enc_mgpu(pk, ec, devices)
    ss2 = kem.dec_mgpu(ct, sk, devices)
    opk, osk = oracle.keypair(k, kc)
    oct_, oss = oracle.enc(k, opk, ec)
    assert np.array_equal(pk, opk) and np.array_equal(sk, osk)
    assert np.array_equal(ct, oct_) and np.array_equal(ss, oss) and np.array_equal(ss2, oss)


def test_multi_device_rejects_absent_device():
    from kyber_amd import lib as kl
    kem = kyber_amd.KyberHip(3)
    with pytest.raises(kl.KyberHipError, match="not present"):
        kem.keypair_mgpu(_coins(4, 64, 1), [0, 63])


@pytest.mark.parametrize("n90", [0, 1])
@pytest.mark.parametrize("n", [1, 300, 2048, 32768, 32769])
def test_small_batch_paths_on_user_streams(oracle, n, n90):
    """Small batches take their own routes through the library (kyber_hip.hip): host calls up to 1 MiB go
    through the pinned mirror on one stream, encapsulations up to 32768 instances expand the matrix on a
    second stream beside the hashes (SHAKE sets up to 65536: in the same launch as the hashes, k_enc_front; key generation runs
    its matrix expansion and PRF as one launch, k_kp_front), and up to 32768 instances the SHAKE-family
    FO hashes run on work-item pairs (keccak_split.h).  Device-pointer calls on two user
    streams in turn, host calls in between: every byte must be the oracle's, on both sides of the
    thresholds, and calls queued back to back on different streams must not disturb each other."""
    import torch
    k = 3
    kem = kyber_amd.KyberHip(k, n90)
    kc, ec, ec2 = _coins(n, 64, 900 + n), _coins(n, 32, 901 + n), _coins(n, 32, 902 + n)
    opk, osk = oracle.keypair(k, kc, n90)
    oct_, oss = oracle.enc(k, opk, ec, n90)
    oct2, oss2 = oracle.enc(k, opk, ec2, n90)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dkc, dec1, dec2 = dev(kc), dev(ec), dev(ec2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        pk, sk = kem.keypair_dev(dkc)
        ct, ss = kem.enc_dev(pk, dec1)
    s2.wait_stream(s1)  # s2 reads pk, sk, ct: data dependencies are the caller's; the workspace is the library's
    with torch.cuda.stream(s2):
        ct2, ssb = kem.enc_dev(pk, dec2)
        ss_dec = kem.dec_dev(ct, sk)
    with torch.cuda.stream(s1):
        s1.wait_stream(s2)
        ss_dec2 = kem.dec_dev(ct2, sk)
    torch.cuda.synchronize()
    assert np.array_equal(pk.cpu().numpy(), opk) and np.array_equal(sk.cpu().numpy(), osk)
    assert np.array_equal(ct.cpu().numpy(), oct_) and np.array_equal(ss.cpu().numpy(), oss)
    assert np.array_equal(ct2.cpu().numpy(), oct2) and np.array_equal(ssb.cpu().numpy(), oss2)
    assert np.array_equal(ss_dec.cpu().numpy(), oss) and np.array_equal(ss_dec2.cpu().numpy(), oss2)
    # host-pointer calls of the same batch (mirror path for the small ones), twice in a row
    for _ in range(2):
        hct, hss = kem.enc(opk, ec)
        assert np.array_equal(hct, oct_) and np.array_equal(hss, oss)
        assert np.array_equal(kem.dec(hct, osk), oss)


@pytest.mark.parametrize("n90", [0, 1])
def test_workspace_calls_in_a_captured_graph(oracle, n90):
    """kyber_batch_enc_dev / dec_dev under stream capture (torch.cuda.CUDAGraph): once the workspace has its size
    (a call of the same batch size outside capture), a captured encapsulation + decapsulation replays with
    the oracle's bytes every time, with eager calls in between.  (The second stream of small encapsulations
    is not used under capture; a captured call that had to GROW the workspace fails instead: no hipMalloc there.)"""
    import torch
    k, n = 3, 100
    kem = kyber_amd.KyberHip(k, n90)
    kc, ec = _coins(n, 64, 7001), _coins(n, 32, 7002)
    opk, osk = oracle.keypair(k, kc, n90)
    oct_, oss = oracle.enc(k, opk, ec, n90)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pk, sk, dec = dev(opk), dev(osk), dev(ec)
    kem.dec_dev(kem.enc_dev(pk, dec)[0], sk)  # warm: workspace sized for n
    torch.cuda.synchronize()
    ct, ss, ssd = (torch.zeros((n, kem.ct_bytes), dtype=torch.uint8, device="cuda"), torch.zeros((n, 32), dtype=torch.uint8, device="cuda"),
                   torch.zeros((n, 32), dtype=torch.uint8, device="cuda"))
    side, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            kem.enc_dev(pk, dec, ct, ss)
            kem.dec_dev(ct, sk, ssd)
    torch.cuda.synchronize()
    assert not bool(ss.any())  # capture records, it does not run
    for _ in range(3):
        ct.zero_(); ss.zero_(); ssd.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(ct.cpu().numpy(), oct_) and np.array_equal(ss.cpu().numpy(), oss) and np.array_equal(ssd.cpu().numpy(), oss)
        c2, s2 = kem.enc_dev(pk, dec)  # an eager call between replays
        torch.cuda.synchronize()
        assert np.array_equal(c2.cpu().numpy(), oct_) and np.array_equal(s2.cpu().numpy(), oss)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_front_kernels_agree_with_the_separate_launches(k):
    """Up to 65536 instances key generation and encapsulation run their independent front stages as one launch
    (k_kp_front, k_enc_front); one instance more and they are the separate kernels of large batches.  The same
    65536 inputs through both: every byte equal (the large-batch path is pinned by the oracle, the vectors and
    the soak)."""
    import torch
    n = 65536
    kem = kyber_amd.KyberHip(k)
    g = torch.Generator(device="cuda")
    g.manual_seed(6500 + k)
    kc = torch.randint(0, 256, (n + 1, 64), dtype=torch.uint8, device="cuda", generator=g)
    ec = torch.randint(0, 256, (n + 1, 32), dtype=torch.uint8, device="cuda", generator=g)
    pk_big, sk_big = kem.keypair_dev(kc)
    ct_big, ss_big = kem.enc_dev(pk_big, ec)
    pk, sk = kem.keypair_dev(kc[:n])
    ct, ss = kem.enc_dev(pk, ec[:n])
    torch.cuda.synchronize()
    assert torch.equal(pk, pk_big[:n]) and torch.equal(sk, sk_big[:n])
    assert torch.equal(ct, ct_big[:n]) and torch.equal(ss, ss_big[:n])
    # decapsulation: up to 65536 instances G, the matrix expansion and H(ct) are one launch and H(ct) waits in the
    # output buffer (k_dec_middle, k2_kdf_dec); valid, corrupted and wrong-key ciphertexts through both routes
    bad = ct_big.clone()
    bad[::3, 5] ^= 0x21
    sk_wrong = sk_big.clone()
    sk_wrong[1::7] = sk_big[0]
    for c_, s_ in ((ct_big, sk_big), (bad, sk_big), (ct_big, sk_wrong)):
        big = kem.dec_dev(c_, s_)
        small = kem.dec_dev(c_[:n].contiguous(), s_[:n].contiguous(), torch.full((n, 32), 0xA5, dtype=torch.uint8, device="cuda"))
        torch.cuda.synchronize()
        assert torch.equal(small, big[:n])
    assert torch.equal(kem.dec_dev(ct, sk), ss)
