"""GPU: the ML-KEM (FIPS 203) mode of libkyber_hip.so (include/mlkem_batch.h)
against the oracle's restatement, bit for bit, for the three parameter sets:
keygen, encaps, decaps, implicit rejection, the input checks of FIPS 203
7.2/7.3, and a 2^16-instance round trip on device pointers.  PARITY UNPINNED
(no ML-KEM vectors exist in the reference or offline; see the header)."""
import numpy as np
import pytest

import kyber_amd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [2, 3, 4])
@pytest.mark.parametrize("n", [1, 9, 300, 2500])
def test_mlkem_matches_oracle(oracle, k, n):
    rng = np.random.default_rng(1000 * k + n)
    kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    mc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ml = kyber_amd.MlKemHip(k)
    ek, dk = ml.keypair(kc)
    oek, odk = oracle.mlkem_keypair(k, kc)
    assert np.array_equal(ek, oek) and np.array_equal(dk, odk)
    ct, ss = ml.encaps(ek, mc)
    oct_, oss = oracle.mlkem_enc(k, oek, mc)
    assert np.array_equal(ct, oct_) and np.array_equal(ss, oss)
    assert np.array_equal(ml.decaps(ct, dk), oss)
    bad = ct.copy()
    bad[::2, rng.integers(0, ct.shape[1])] ^= 0x04
    got = ml.decaps(bad, dk)
    assert np.array_equal(got, oracle.mlkem_dec(k, bad, odk))
    assert not np.any(np.all(got[::2] == ss[::2], axis=1))
    assert np.array_equal(got[1::2], ss[1::2])


def test_mlkem_first_instance_matches_python_model():
    """One instance per set straight against the standard's algorithms on Python integers."""
    import mlkem_model as mm
    for k in (2, 3, 4):
        d, z, m = bytes(range(32)), bytes(range(32, 64)), bytes(range(64, 96))
        ml = kyber_amd.MlKemHip(k)
        ek, dk = ml.keypair(np.frombuffer(d + z, np.uint8))
        mek, mdk = mm.keygen_internal(k, d, z)
        assert bytes(ek[0]) == mek and bytes(dk[0]) == mdk
        ct, ss = ml.encaps(ek, np.frombuffer(m, np.uint8))
        K, c = mm.encaps_internal(k, mek, m)
        assert bytes(ct[0]) == c and bytes(ss[0]) == K


@pytest.mark.parametrize("k", [2, 3, 4])
def test_mlkem_input_checks(oracle, k):
    rng = np.random.default_rng(55 + k)
    n = 40
    ml = kyber_amd.MlKemHip(k)
    ek, dk = ml.keypair(rng.integers(0, 256, (n, 64), dtype=np.uint8))
    assert not ml.check_ek(ek).any() and not ml.check_dk(dk).any()
    bek = ek.copy()
    bek[3, 0:2] = 0xFF          # coefficient 0 = 0xFFF >= q
    bek[7, 384 * k - 2:384 * k] = 0xFF   # last coefficient
    f = ml.check_ek(bek)
    assert f[3] == 1 and f[7] == 1 and f.sum() == 2
    bdk = dk.copy()
    bdk[5, 768 * k + 40] ^= 1   # stored H(ek)
    bdk[9, 384 * k + 1] ^= 0x80  # a byte of the embedded ek (coefficient stays < q with high probability -> hash mismatch)
    bdk[11, 384 * k:384 * k + 2] = 0xFF
    f = ml.check_dk(bdk)
    assert f[5] == 2 and (f[9] & 2) and (f[11] & 1) and (f[11] & 2)
    assert not np.delete(f, [5, 9, 11]).any()


def test_mlkem_device_round_trip_65536():
    import torch
    k, n = 3, 1 << 16
    ml = kyber_amd.MlKemHip(k)
    g = torch.Generator(device="cuda")
    g.manual_seed(203)
    kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    mc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    ek, dk = ml.keypair_dev(kc)
    ct, ss = ml.encaps_dev(ek, mc)
    assert torch.equal(ml.decaps_dev(ct, dk), ss)
    ct[::4, 9] ^= 0x20
    r = ml.decaps_dev(ct, dk)
    same = (r == ss).all(dim=1)
    assert not same[::4].any() and same[1::4].all() and same[2::4].all() and same[3::4].all()


@pytest.mark.parametrize("k", [2, 3, 4])
def test_mlkem_small_batch_routes_agree_with_the_separate_launches(oracle, k):
    """Batches up to 65 536 run the front / middle launches on work-item pairs (keccak_split.h), larger ones the
    separate kernels: the first 65 536 instances of a 65 537-instance call give the same bytes as a call of
    65 536, rejected ciphertexts and wrong keys included; a strided sample is checked against the oracle."""
    import torch
    n = 65536
    ml = kyber_amd.MlKemHip(k)
    g = torch.Generator(device="cuda")
    g.manual_seed(2030 + k)
    kc = torch.randint(0, 256, (n + 1, 64), dtype=torch.uint8, device="cuda", generator=g)
    mc = torch.randint(0, 256, (n + 1, 32), dtype=torch.uint8, device="cuda", generator=g)
    ek_b, dk_b = ml.keypair_dev(kc)                 # separate launches
    ek_s, dk_s = ml.keypair_dev(kc[:n].contiguous())  # small-batch routes
    assert torch.equal(ek_s, ek_b[:n]) and torch.equal(dk_s, dk_b[:n])
    ct_b, ss_b = ml.encaps_dev(ek_b, mc)
    ct_s, ss_s = ml.encaps_dev(ek_s, mc[:n].contiguous())
    assert torch.equal(ct_s, ct_b[:n]) and torch.equal(ss_s, ss_b[:n])
    bad = ct_b.clone()
    bad[::3, 11] ^= 0x02                            # every third ciphertext is rejected
    dk_w = dk_b.clone()
    dk_w[1::3, 5] ^= 0x01                           # every third key (offset 1) has another secret vector
    r_b = ml.decaps_dev(bad, dk_w)
    r_s = ml.decaps_dev(bad[:n].contiguous(), dk_w[:n].contiguous())
    assert torch.equal(r_s, r_b[:n])
    same = (r_s == ss_s).all(dim=1)
    assert not same[::3].any() and same[2::3].all()
    idx = torch.arange(0, n, 1021, device="cuda")
    oek, odk = oracle.mlkem_keypair(k, kc[idx].cpu().numpy())
    assert np.array_equal(ek_s[idx].cpu().numpy(), oek) and np.array_equal(dk_s[idx].cpu().numpy(), odk)
    oct_, oss = oracle.mlkem_enc(k, oek, mc[idx].cpu().numpy())
    assert np.array_equal(ct_s[idx].cpu().numpy(), oct_) and np.array_equal(ss_s[idx].cpu().numpy(), oss)
    assert np.array_equal(r_s[idx].cpu().numpy(), oracle.mlkem_dec(k, bad[idx].cpu().numpy(), dk_w[idx].cpu().numpy()))


@pytest.mark.parametrize("n", [1, 777, 32768, 32769])
def test_mlkem_small_batches_on_a_user_stream_and_in_a_graph(oracle, n):
    """The small-batch routes on a caller's stream, eagerly and as a captured graph replayed three times
    (one stream, kernels only: the calls are capturable), against the oracle on a strided sample."""
    import torch
    k = 3
    ml = kyber_amd.MlKemHip(k)
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    mc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    idx = torch.arange(0, n, max(1, n // 40), device="cuda")
    oek, odk = oracle.mlkem_keypair(k, kc[idx].cpu().numpy())
    oct_, oss = oracle.mlkem_enc(k, oek, mc[idx].cpu().numpy())
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ek, dk = ml.keypair_dev(kc)
        ct, ss = ml.encaps_dev(ek, mc)
        ss2 = ml.decaps_dev(ct, dk)
    st.synchronize()
    assert np.array_equal(ek[idx].cpu().numpy(), oek) and np.array_equal(dk[idx].cpu().numpy(), odk)
    assert np.array_equal(ct[idx].cpu().numpy(), oct_) and np.array_equal(ss[idx].cpu().numpy(), oss)
    assert torch.equal(ss, ss2)
    ml.decaps_dev(ct, dk)   # the workspace is laid out before the capture starts
    torch.cuda.synchronize()
    import ctypes
    L = kyber_amd.lib.library()
    out_ct, out_ss, out_dec = torch.zeros_like(ct), torch.zeros_like(ss), torch.zeros_like(ss)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(graph, stream=st):
            sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            assert L.mlkem_batch_encaps_dev(k, n, p(out_ct), p(out_ss), p(ek), p(mc), sp) == 0
            assert L.mlkem_batch_decaps_dev(k, n, p(out_dec), p(out_ct), p(dk), sp) == 0
    for _ in range(3):
        out_ct.zero_(), out_ss.zero_(), out_dec.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out_ct, ct) and torch.equal(out_ss, ss) and torch.equal(out_dec, ss)
