"""GPU: single kernels against the oracle's stage functions, bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

import kyber_amd
from kyber_amd import lib as kl

pytestmark = pytest.mark.gpu
Q = 3329


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_gen_matrix_90s_matches_oracle(oracle):
    """AES-256-CTR XOF (ref/symmetric-aes.c:7-13): continuous stream parse with
    the reference's 2-byte carry between its 64-byte refills."""
    for k in (2, 3, 4):
        n = 40
        seeds = np.random.default_rng(90 + k).integers(0, 256, (n, 32), dtype=np.uint8)
        for tr in (0, 1):
            got = kyber_amd.KyberHip(k, 1).gen_matrix_dev(_dev(seeds), tr).cpu().numpy()
            for i in range(n):
                assert np.array_equal(got[i], oracle.gen_matrix(k, seeds[i], tr, 1)[0]), (k, tr, i)


@pytest.mark.parametrize("k", [2, 3, 4])
@pytest.mark.parametrize("transposed", [0, 1])
def test_gen_matrix_matches_oracle(oracle, k, transposed):
    """ref/indcpa.c:164-192, including seeds whose entries need a 4th SHAKE128
    block (P = 0.8 % per entry, so 300 seeds hold dozens of them)."""
    n = 300
    seeds = np.random.default_rng(7 * k + transposed).integers(0, 256, (n, 32), dtype=np.uint8)
    got = kyber_amd.KyberHip(k).gen_matrix_dev(_dev(seeds), transposed).cpu().numpy()
    extra = 0
    for i in range(n):
        want, nb = oracle.gen_matrix(k, seeds[i], transposed)
        extra += int((nb > 3).sum())
        assert np.array_equal(got[i], want), (i, nb)
    assert extra > 0, "no entry exercised the 4th-block path"
    assert got.min() >= 0 and got.max() < Q


def test_gen_matrix_ragged_job_count(oracle):
    """job counts that do not fill the last wavefront/workgroup"""
    for n in (1, 3, 29):
        seeds = np.random.default_rng(n).integers(0, 256, (n, 32), dtype=np.uint8)
        got = kyber_amd.KyberHip(3).gen_matrix_dev(_dev(seeds), 1).cpu().numpy()
        for i in range(n):
            assert np.array_equal(got[i], oracle.gen_matrix(3, seeds[i], 1)[0])


def test_poly_ntt_wide_inputs_and_extremes(oracle):
    """poly_ntt on inputs up to 2q in magnitude (the widest for which the
    reference's int16 butterflies cannot overflow: 2q + 7q < 2^15) and on the
    constant extremes; the output must be the Barrett-centred representative."""
    rng = np.random.default_rng(99)
    a = rng.integers(-2 * Q, 2 * Q + 1, (257, 256)).astype(np.int16)
    a[0, :] = 2 * Q
    a[1, :] = -2 * Q
    a[2, :] = 0
    a[3, ::2], a[3, 1::2] = Q - 1, -(Q - 1)
    got = kl.poly_ntt_(_dev(a)).cpu().numpy()
    want = oracle.poly_op("ko_poly_ntt", a)
    assert np.array_equal(got, want)
    assert got.min() >= -(Q - 1) // 2 and got.max() <= (Q - 1) // 2


@pytest.mark.parametrize("npoly", [1, 8, 13, 1000])
def test_poly_ntt_invntt_basemul(oracle, npoly):
    rng = np.random.default_rng(npoly)
    a = rng.integers(0, Q, (npoly, 256)).astype(np.int16)
    b = rng.integers(-Q // 2, Q // 2 + 1, (npoly, 256)).astype(np.int16)
    # poly_ntt (ref/poly.c:244-248): exact representatives, not just mod q
    ta = kl.poly_ntt_(_dev(a)).cpu().numpy()
    want_a = oracle.poly_op("ko_poly_ntt", a)
    assert np.array_equal(ta, want_a)
    tb = kl.poly_ntt_(_dev(b)).cpu().numpy()
    want_b = oracle.poly_op("ko_poly_ntt", b)
    assert np.array_equal(tb, want_b)
    # basemul (ref/poly.c:273-280)
    r = torch.empty((npoly, 256), dtype=torch.int16, device="cuda")
    prod = kl.poly_basemul(r, _dev(want_a), _dev(want_b)).cpu().numpy()
    want_p = oracle.poly_basemul(want_a, want_b)
    assert np.array_equal(prod, want_p)
    # invntt_tomont (ref/ntt.c:106-126) on Barrett-reduced input, as the KEM feeds it
    red = np.array([[oracle.lib.ko_barrett_reduce(int(v)) for v in row] for row in want_p[:min(npoly, 32)]], np.int16)
    inv = kl.poly_invntt_(_dev(red)).cpu().numpy()
    assert np.array_equal(inv, oracle.poly_op("ko_invntt_tomont", red))


def test_poly_mul_fused_is_negacyclic_product(oracle):
    """fused ntt -> basemul -> invntt equals the schoolbook product in
    Z_q[X]/(X^256+1) (times the Montgomery factor the reference also carries)."""
    npoly = 64
    rng = np.random.default_rng(3)
    a = rng.integers(0, Q, (npoly, 256)).astype(np.int16)
    b = rng.integers(0, Q, (npoly, 256)).astype(np.int16)
    r = torch.empty((npoly, 256), dtype=torch.int16, device="cuda")
    got = kl.poly_mul_fused(r, _dev(a), _dev(b)).cpu().numpy().astype(np.int64)
    # oracle pipeline, exact
    ta, tb = oracle.poly_op("ko_poly_ntt", a), oracle.poly_op("ko_poly_ntt", b)
    p = oracle.poly_basemul(ta, tb)
    p = np.array([[oracle.lib.ko_barrett_reduce(int(v)) for v in row] for row in p], np.int16)
    assert np.array_equal(got, oracle.poly_op("ko_invntt_tomont", p).astype(np.int64))
    # algebra: basemul carries 2^-16, invntt_tomont carries 2^16 -> plain product mod q
    for i in range(4):
        full = np.convolve(a[i].astype(np.int64), b[i].astype(np.int64))
        neg = full[:256].copy()
        neg[:255] -= full[256:]
        assert np.array_equal(np.mod(got[i], Q), np.mod(neg, Q))


def test_device_pointer_api_and_chunking(oracle):
    """*_dev entry points on torch tensors, with a chunk size that forces
    several internal passes and a ragged last one."""
    k, n = 3, 1000
    kem = kyber_amd.KyberHip(k)
    rng = np.random.default_rng(11)
    kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    try:
        kl.set_chunk(384)
        pk, sk = kem.keypair_dev(_dev(kc))
        ct, ss = kem.enc_dev(pk, _dev(ec))
        ss2 = kem.dec_dev(ct, sk)
        torch.cuda.synchronize()
    finally:
        kl.set_chunk(0)
    opk, osk = oracle.keypair(k, kc)
    oct_, oss = oracle.enc(k, opk, ec)
    assert np.array_equal(pk.cpu().numpy(), opk) and np.array_equal(sk.cpu().numpy(), osk)
    assert np.array_equal(ct.cpu().numpy(), oct_) and np.array_equal(ss.cpu().numpy(), oss)
    assert np.array_equal(ss2.cpu().numpy(), oss)


def test_namespaced_single_call_api(oracle):
    """pqcrystals_kyber768_hip_{keypair,enc,dec} (ref/api.h:39-41): a batch of one,
    entropy from randombytes(); the round trip must agree and match the oracle's
    view of the same pk/sk/ct."""
    lib = kyber_amd.library()
    pkb, skb, ctb = kyber_amd.sizes(3)
    pk, sk = (ctypes.c_uint8 * pkb)(), (ctypes.c_uint8 * skb)()
    ct, ss, ss2 = (ctypes.c_uint8 * ctb)(), (ctypes.c_uint8 * 32)(), (ctypes.c_uint8 * 32)()
    assert lib.pqcrystals_kyber768_hip_keypair(pk, sk) == 0
    assert lib.pqcrystals_kyber768_hip_enc(ct, ss, pk) == 0
    assert lib.pqcrystals_kyber768_hip_dec(ss2, ct, sk) == 0
    assert bytes(ss) == bytes(ss2) and any(bytes(ss))
    want = oracle.dec(3, np.frombuffer(bytes(ct), np.uint8), np.frombuffer(bytes(sk), np.uint8))
    assert bytes(want[0]) == bytes(ss)
    assert bytes(sk)[384 * 3:384 * 3 + pkb] == bytes(pk)  # sk embeds pk (ref/kem.c:28-29)


def test_release_and_profile_housekeeping(oracle):
    """kyber_hip_release drops the workspace, streams and staging; the next call
    rebuilds them.  The profiling hook reports every kernel of a call."""
    kem = kyber_amd.KyberHip(3)
    kc = np.random.default_rng(8).integers(0, 256, (50, 64), dtype=np.uint8)
    pk0, sk0 = kem.keypair(kc)
    assert kl.library().kyber_hip_release() == 0
    assert kl.library().kyber_hip_release() == 0          # idempotent
    kl.profile_enable(True)
    pk1, sk1 = kem.keypair(kc)
    prof = kl.profile_read()
    kl.profile_enable(False)
    assert np.array_equal(pk0, pk1) and np.array_equal(sk0, sk1)
    # 50 instances: the matrix expansion and the PRF are one launch (k_kp_front, keccak_split.h)
    assert {"k_kp_pre", "k_kp_front", "k_sample_matrix_tail", "k_kp_arith", "k_kp_post"} <= set(prof)
    assert all(cnt == 1 and ms >= 0 for cnt, ms in prof.values())
