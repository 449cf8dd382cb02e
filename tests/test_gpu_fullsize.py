"""GPU: BASELINE.json's full sizes through size-independent properties.
2^20 Kyber768 instances: decaps(encaps) round trip for every instance, a
strided sample bit-exact against the oracle, implicit rejection on a corrupted
subset; 2^20-polynomial NTT round trip."""
import numpy as np
import pytest
import torch

import kyber_amd
from kyber_amd import lib as kl

pytestmark = pytest.mark.gpu


def test_kyber768_one_million_roundtrip(oracle):
    k, n = 3, 1 << 20
    kem = kyber_amd.KyberHip(k)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x4B59424552)
    kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    pk, sk = kem.keypair_dev(kc)
    ct, ssb = kem.enc_dev(pk, ec)
    ssa = kem.dec_dev(ct, sk)
    assert torch.equal(ssa, ssb)
    assert torch.equal(sk[:, 384 * k:384 * k + kem.pk_bytes], pk)          # sk embeds pk
    assert torch.equal(sk[:, -32:], kc[:, 32:])                            # z
    # distinct instances give distinct secrets (sorted-unique over a sample)
    s = ssb[:: 64].cpu().numpy()
    assert len({bytes(r) for r in s}) == len(s)
    idx = torch.arange(0, n, 4099, device="cuda")
    opk, osk = oracle.keypair(k, kc[idx].cpu().numpy())
    oct_, oss = oracle.enc(k, opk, ec[idx].cpu().numpy())
    assert np.array_equal(pk[idx].cpu().numpy(), opk) and np.array_equal(sk[idx].cpu().numpy(), osk)
    assert np.array_equal(ct[idx].cpu().numpy(), oct_) and np.array_equal(ssb[idx].cpu().numpy(), oss)
    # corrupt every 8th ciphertext: those, and only those, must reject
    bad = ct.clone()
    bad[::8, 17] ^= 0x01
    ssr = kem.dec_dev(bad, sk)
    same = (ssr == ssb).all(dim=1)
    assert not same[::8].any()
    mask = torch.ones(n, dtype=torch.bool, device="cuda")
    mask[::8] = False
    assert same[mask].all()
    j = torch.arange(0, n, 8 * 1031, device="cuda")
    assert np.array_equal(ssr[j].cpu().numpy(), oracle.dec(k, bad[j].cpu().numpy(), sk[j].cpu().numpy()))


def test_ntt_roundtrip_one_million_polys():
    """invntt_tomont(poly_ntt(a)) == a * 2^16 mod q for 2^20 polynomials."""
    n, Q = 1 << 20, 3329
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    a = torch.randint(0, Q, (n, 256), dtype=torch.int16, device="cuda", generator=g)
    r = a.clone()
    kl.poly_ntt_(r)
    assert int(r.abs().max()) <= Q // 2
    kl.poly_invntt_(r)
    want = (a.to(torch.int32) * ((1 << 16) % Q)) % Q
    assert torch.equal(r.to(torch.int32) % Q, want)


def test_host_pipeline_multi_pass_pinned_and_pageable(oracle):
    """Host-pointer entry points on a batch that spans several copy/compute
    passes (HOST_PASS = 2^15, ragged last pass), from pinned and from pageable
    memory: same bytes as the device-pointer path, strided sample vs the oracle."""
    import torch
    from kyber_amd import lib as kl
    k, n = 3, 3 * 32768 + 777
    kem = kyber_amd.KyberHip(k)
    rng = np.random.default_rng(2718)
    kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = kem.keypair(kc)
    ct, ss = kem.enc(pk, ec)
    ss2 = kem.dec(ct, sk)
    assert np.array_equal(ss, ss2)
    dev = torch.device("cuda", 0)
    dpk, dsk = kem.keypair_dev(torch.from_numpy(kc).to(dev))
    dct, dss = kem.enc_dev(dpk, torch.from_numpy(ec).to(dev))
    assert np.array_equal(dpk.cpu().numpy(), pk) and np.array_equal(dsk.cpu().numpy(), sk)
    assert np.array_equal(dct.cpu().numpy(), ct) and np.array_equal(dss.cpu().numpy(), ss)
    idx = np.arange(0, n, 1571)
    opk, osk = oracle.keypair(k, kc[idx])
    oct_, oss = oracle.enc(k, opk, ec[idx])
    assert np.array_equal(pk[idx], opk) and np.array_equal(sk[idx], osk)
    assert np.array_equal(ct[idx], oct_) and np.array_equal(ss[idx], oss)
    # pinned buffers, shared-key variants across passes
    ppk, pec = kl.pinned_empty(pk.shape), kl.pinned_empty(ec.shape)
    ppk[:], pec[:] = pk, ec
    pct, pss = kem.enc(ppk, pec)
    assert np.array_equal(pct, ct) and np.array_equal(pss, ss)
    sct, sss = kem.enc_shared_pk(pk[5], ec)
    assert np.array_equal(kem.dec_shared_sk(sct, sk[5]), sss)
    assert np.array_equal(sct[idx], oracle.enc(k, np.tile(pk[5], (len(idx), 1)), ec[idx])[0])
    del ppk, pec


def test_calls_on_different_streams_do_not_race_on_the_workspace():
    """Two encapsulation batches queued on two streams without any host
    synchronisation in between: the library orders them on the device (one
    workspace per device), so both results equal their single-stream runs."""
    kem = kyber_amd.KyberHip(3)
    dev = torch.device("cuda", 0)
    n = 1 << 16
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device=dev, generator=g)
    e1 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    e2 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    pk, sk = kem.keypair_dev(kc)
    want1, want_s1 = kem.enc_dev(pk, e1)
    want2, want_s2 = kem.enc_dev(pk, e2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            ct1, ss1 = kem.enc_dev(pk, e1)
        with torch.cuda.stream(s2):
            ct2, ss2 = kem.enc_dev(pk, e2)
            back = kem.dec_dev(ct2, sk)
        torch.cuda.synchronize()
        assert torch.equal(ct1, want1) and torch.equal(ss1, want_s1)
        assert torch.equal(ct2, want2) and torch.equal(ss2, want_s2) and torch.equal(back, want_s2)


def test_polynomial_kernel_properties_at_full_size():
    """2^20 polynomials: linearity of the NTT mod q, commutativity of the base
    multiplication mod q, and the fused product equal to the three separate
    kernels bit for bit (ntt, ntt, basemul, Barrett, invntt_tomont)."""
    n, Q = 1 << 20, 3329
    g = torch.Generator(device="cuda")
    g.manual_seed(2025)
    a = torch.randint(0, Q, (n, 256), dtype=torch.int16, device="cuda", generator=g)
    b = torch.randint(-(Q // 2), Q // 2 + 1, (n, 256), dtype=torch.int16, device="cuda", generator=g)
    s = ((a.to(torch.int32) + b.to(torch.int32)) % Q).to(torch.int16)
    ta, tb, ts = kl.poly_ntt_(a.clone()), kl.poly_ntt_(b.clone()), kl.poly_ntt_(s)
    assert torch.equal((ta.to(torch.int32) + tb.to(torch.int32)) % Q, ts.to(torch.int32) % Q)
    ab = kl.poly_basemul(torch.empty_like(a), ta, tb)
    ba = kl.poly_basemul(torch.empty_like(a), tb, ta)
    assert torch.equal(ab.to(torch.int32) % Q, ba.to(torch.int32) % Q)
    fused = kl.poly_mul_fused(torch.empty_like(a), a, b)
    # the separate path: Barrett-reduce the product (values are in (-2q, 2q)) exactly as poly_reduce does
    red = ab.to(torch.int32)
    t = (20159 * red + (1 << 25)) >> 26
    red = (red - t * Q).to(torch.int16)
    assert torch.equal(kl.poly_invntt_(red), fused)


@pytest.mark.parametrize("n", [40, 3000])
def test_concurrent_host_threads_on_one_device(oracle, n):
    """Four host threads call the host-pointer API at once (ctypes releases the
    GIL): the per-device locks serialise them, every result is the
    single-threaded one.  n = 40 goes through the pinned mirror, the work-item-pair
    hashes and the second stream of small encapsulations (one of each per device),
    n = 3000 through the pipelined copies."""
    import threading
    kem = kyber_amd.KyberHip(2)
    rng = np.random.default_rng(4)
    kcs = [rng.integers(0, 256, (n, 64), dtype=np.uint8) for _ in range(4)]
    ecs = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for _ in range(4)]
    want = []
    for kc, ec in zip(kcs, ecs):
        pk, sk = kem.keypair(kc)
        ct, ss = kem.enc(pk, ec)
        want.append((pk, sk, ct, ss))
    got = [None] * 4

    def work(i):
        for _ in range(8 if n < 100 else 1):
            pk, sk = kem.keypair(kcs[i])
            ct, ss = kem.enc(pk, ecs[i])
            got[i] = (pk, sk, ct, ss, kem.dec(ct, sk))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(4):
        for a, b in zip(got[i][:4], want[i]):
            assert np.array_equal(a, b)
        assert np.array_equal(got[i][4], want[i][3])
    idx = np.arange(0, n, 500 if n > 500 else 7)
    assert np.array_equal(want[0][0][idx], oracle.keypair(2, kcs[0][idx])[0])
