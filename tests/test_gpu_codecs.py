"""GPU: the single-function stages of include/kyber_batch.h -- noise (PRF + centred binomial
map), the 12-bit and d-bit codecs, message encoding, and the accumulated base multiplication --
each against the oracle's restatement of the reference function (a7, a11, a13-a17).
The kernels are built from the device functions the KEM kernels use, so a KEM parity failure
can be localised here."""
import numpy as np
import pytest
import torch

from kyber_amd import lib as kl

pytestmark = pytest.mark.gpu
Q = 3329


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("k,n90", [(2, 0), (3, 0), (4, 0), (2, 1), (3, 1)])
def test_getnoise_matches_oracle(oracle, k, n90):
    """ref/poly.c:208-232: eta1 (3 for Kyber512) below nonce K, eta2 = 2 from there, as indcpa_enc draws them."""
    rng = np.random.default_rng(50 + k + n90)
    n, nnonce = 37, 2 * k + 1
    seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    got = kl.getnoise(k, n90, _dev(seeds), nnonce, k).cpu().numpy()
    eta1 = 3 if k == 2 else 2
    for i in range(0, n, 5):
        for nonce in range(nnonce):
            want = oracle.getnoise(k, n90, eta1 if nonce < k else 2, seeds[i], nonce)
            assert np.array_equal(got[i, nonce], want), (i, nonce)
    # key generation's use: every nonce with eta1 (ref/indcpa.c:223-226)
    got = kl.getnoise(k, n90, _dev(seeds), 2 * k, 2 * k).cpu().numpy()
    assert np.array_equal(got[3, 2 * k - 1], oracle.getnoise(k, n90, eta1, seeds[3], 2 * k - 1))


@pytest.mark.parametrize("eta", [2, 3])
def test_cbd_matches_oracle_including_extremes(oracle, eta):
    rng = np.random.default_rng(eta)
    buf = rng.integers(0, 256, (101, 64 * eta), dtype=np.uint8)
    buf[0], buf[1], buf[2] = 0x00, 0xFF, 0x55
    got = kl.poly_cbd(eta, _dev(buf)).cpu().numpy()
    assert np.array_equal(got, oracle.cbd(eta, buf))
    assert got.min() >= -eta and got.max() <= eta


def test_tobytes_frombytes_match_oracle(oracle):
    rng = np.random.default_rng(7)
    a = rng.integers(-(Q - 1), Q, (67, 256)).astype(np.int16)      # any representative in (-q, q): ref/poly.c:118-121
    a[0], a[1], a[2] = Q - 1, -(Q - 1), 0
    got = kl.poly_tobytes(_dev(a)).cpu().numpy()
    assert np.array_equal(got, oracle.codec("ko_poly_tobytes", a, (384,), np.uint8))
    b = rng.integers(0, 256, (67, 384), dtype=np.uint8)            # frombytes masks 12 bits and nothing else: values up to 4095
    b[0] = 0xFF
    got = kl.poly_frombytes(_dev(b)).cpu().numpy()
    assert np.array_equal(got, oracle.codec("ko_poly_frombytes", b, (256,), np.int16))
    assert got.max() == 4095


@pytest.mark.parametrize("d", [4, 5, 10, 11])
def test_compress_decompress_match_oracle(oracle, d):
    rng = np.random.default_rng(d)
    a = rng.integers(-(Q - 1), Q, (75, 256)).astype(np.int16)
    a[0] = np.arange(256) * 13 % Q
    a[1] = -(np.arange(256) * 13 % Q)
    a[2, :] = np.resize(np.array([1664, 1665, -1664, -1665, Q - 1, 1 - Q, 0, 1], np.int16), 256)
    got = kl.poly_compress(d, _dev(a)).cpu().numpy()
    assert np.array_equal(got, oracle.codec("ko_poly_compress", a, (32 * d,), np.uint8, d))
    b = rng.integers(0, 256, (75, 32 * d), dtype=np.uint8)
    b[0], b[1] = 0xFF, 0x00
    got = kl.poly_decompress(d, _dev(b)).cpu().numpy()
    assert np.array_equal(got, oracle.codec("ko_poly_decompress", b, (256,), np.int16, d))


def test_every_residue_compresses_like_the_reference(oracle):
    """all 2q - 1 representatives in (-q, q), every d: the one-FMA compression of fpoly.h on the GPU itself"""
    vals = np.arange(-(Q - 1), Q, dtype=np.int16)
    a = np.resize(vals, (27, 256)).astype(np.int16)
    for d in (1, 4, 5, 10, 11):
        got = kl.poly_compress(d, _dev(a)).cpu().numpy()
        if d == 1:
            want = oracle.codec("ko_poly_tomsg", a, (32,), np.uint8)
        else:
            want = oracle.codec("ko_poly_compress", a, (32 * d,), np.uint8, d)
        assert np.array_equal(got, want), d


def test_message_encoding_matches_oracle(oracle):
    rng = np.random.default_rng(9)
    msg = rng.integers(0, 256, (41, 32), dtype=np.uint8)
    msg[0], msg[1] = 0xFF, 0x00
    got = kl.poly_decompress(1, _dev(msg)).cpu().numpy()                 # poly_frommsg
    assert np.array_equal(got, oracle.codec("ko_poly_frommsg", msg, (256,), np.int16))
    a = rng.integers(-(Q - 1), Q, (41, 256)).astype(np.int16)
    a[0, :] = np.resize(np.array([832, 833, 2496, 2497, -832, -833, -2496, -2497], np.int16), 256)  # the rounding boundaries q/4, 3q/4
    got = kl.poly_compress(1, _dev(a)).cpu().numpy()                     # poly_tomsg
    assert np.array_equal(got, oracle.codec("ko_poly_tomsg", a, (32,), np.uint8))


@pytest.mark.parametrize("k", [2, 3, 4])
def test_polyvec_basemul_acc_matches_oracle_mod_q(oracle, k):
    """The fp32 accumulation of k_enc_arith (cached multiplicand, unreduced FMA chains, offset
    correction) against ref/polyvec.c:189-201: the reference's result carries 2^-16 (Montgomery
    products), the kernel's does not, so r_ref * 2^16 = r_gpu mod q."""
    rng = np.random.default_rng(30 + k)
    nvec = 53
    a = rng.integers(0, Q, (nvec, k, 256)).astype(np.int16)               # canonical, as the matrix sampler writes them
    b = rng.integers(-1664, 1665, (nvec, k, 256)).astype(np.int16)        # centred, as an NTT leaves them
    a[0], b[0] = Q - 1, 1664                                               # the accumulation bound of fpoly.h, every term positive
    a[1], b[1] = Q - 1, -1664
    a[2, :, 0::2], a[2, :, 1::2], b[2] = Q - 1, 0, 1664
    got = kl.polyvec_basemul_acc(k, _dev(a), _dev(b)).cpu().numpy().astype(np.int64)
    want = oracle.polyvec_basemul_acc(k, a, b).astype(np.int64)
    assert np.abs(got).max() <= 1665
    assert np.all((want * 65536 - got) % Q == 0)


def test_stage_entry_points_reject_bad_arguments():
    import ctypes
    L = kl.library()
    x = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    p = ctypes.c_void_p(x.data_ptr())
    assert L.kyber_batch_poly_cbd_dev(4, 1, p, p, None) == -1
    assert L.kyber_batch_poly_compress_dev(3, 1, p, p, None) == -1
    assert L.kyber_batch_poly_decompress_dev(12, 1, p, p, None) == -1
    assert L.kyber_batch_polyvec_basemul_acc_dev(5, 1, p, p, p, None) == -1
    assert L.kyber_batch_getnoise_dev(3, 0, 1, 0, 0, p, p, None) == -1
    assert L.kyber_batch_poly_tobytes_dev(0, p, p, None) == 0
    assert L.kyber_batch_poly_frombytes_dev(1, None, p, None) == -1


def test_keccak_f1600_matches_oracle(oracle):
    """ref/fips202.c:82-344 on the GPU's (lo, hi) halves with the phase-locked schedule: zero state
    (the published first lane F1258F7940E1DDE7), all-ones, random; a ragged count of states."""
    import ctypes
    rng = np.random.default_rng(1600)
    n = 1000
    st = rng.integers(0, 1 << 63, (n, 25), dtype=np.int64) * 2 + rng.integers(0, 2, (n, 25), dtype=np.int64)
    st[0] = 0
    st[1] = -1
    want = st.copy()
    for i in list(range(8)) + list(range(8, n, 97)) + [n - 1]:
        row = np.ascontiguousarray(want[i])
        oracle.lib.ko_keccak_f1600(ctypes.c_void_p(row.ctypes.data))
        want[i] = row
    got = kl.keccak_f1600_(_dev(st)).cpu().numpy()
    assert got[0, 0] == np.int64(np.uint64(0xF1258F7940E1DDE7).astype(np.int64))
    for i in list(range(8)) + list(range(8, n, 97)) + [n - 1]:
        assert np.array_equal(got[i], want[i]), i


@pytest.mark.parametrize("k,n90", [(2, 0), (3, 0), (4, 0), (3, 1), (4, 1)])
def test_hash_h_of_keys_and_ciphertexts_matches_oracle(oracle, k, n90):
    """H = sha3_256 (sha256 for the 90s sets) over pk- and ct-sized inputs: the multi-block absorb
    loops of the FO kernels (with their next-block prefetch) on their own."""
    import ctypes
    p = oracle.params(k, n90)
    rng = np.random.default_rng(k + 10 * n90)
    fn = oracle.lib.ko_sha256 if n90 else oracle.lib.ko_sha3_256
    for of_ct, size in ((0, oracle.lib.ko_pk_bytes(ctypes.byref(p))), (1, oracle.lib.ko_ct_bytes(ctypes.byref(p)))):
        n = 77
        data = rng.integers(0, 256, (n, size), dtype=np.uint8)
        data[0], data[1] = 0x00, 0xFF
        got = kl.hash_h(k, n90, _dev(data), of_ct).cpu().numpy()
        for i in range(0, n, 4):
            h = np.empty(32, np.uint8)
            row = np.ascontiguousarray(data[i])
            fn(ctypes.c_void_p(h.ctypes.data), ctypes.c_void_p(row.ctypes.data), size)
            assert np.array_equal(got[i], h), (of_ct, i)


@pytest.mark.parametrize("n90", [0, 1])
@pytest.mark.parametrize("inlen", [32, 64])
def test_hash_g_matches_oracle(oracle, n90, inlen):
    """G = sha3_512 (sha512 for the 90s sets, ref/sha512.c:77-302) on the two input lengths the KEM
    uses: 32 bytes (G(d), key generation) and 64 bytes (G(m || H(pk)), encapsulation and decapsulation)"""
    import ctypes
    n = 1000
    data = np.random.default_rng(inlen + n90).integers(0, 256, (n, inlen), dtype=np.uint8)
    data[0], data[1], data[2, -1] = 0x00, 0xFF, 0x80
    got = kl.hash_g(n90, _dev(data)).cpu().numpy()
    fn = oracle.lib.ko_sha512 if n90 else oracle.lib.ko_sha3_512
    for i in range(n):
        h = np.empty(64, np.uint8)
        row = np.ascontiguousarray(data[i])
        fn(ctypes.c_void_p(h.ctypes.data), ctypes.c_void_p(row.ctypes.data), inlen)
        assert np.array_equal(got[i], h), i


@pytest.mark.parametrize("first,nblocks", [(0, 8), (0, 33), (250, 12), (255, 2), (256, 5), (0xFFFFFFF0, 4)])
def test_aes256ctr_keystream_matches_oracle(oracle, first, nblocks):
    """The 90s sets' XOF / PRF keystream on its own, also past block 255, where the folded first two
    rounds of the counter-mode kernels give way to full rounds (a matrix entry needs about 30 blocks,
    so the KEM tests never get there)."""
    import ctypes
    import torch
    rng = np.random.default_rng(first % 1000 + nblocks)
    for n in (300, 1):
        keys = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        nonces = rng.integers(0, 1 << 16, n, dtype=np.uint16)
        keys[0] = 0
        nonces[0] = 0
        got = kl.aes256ctr(_dev(keys), torch.from_numpy(nonces.view(np.int16)).cuda(), nblocks, first).cpu().numpy()
        for i in range(0, n, 7):
            want = np.empty(nblocks * 16, np.uint8)
            nonce = np.zeros(12, np.uint8)
            nonce[0], nonce[1] = nonces[i] & 0xFF, nonces[i] >> 8
            key = np.ascontiguousarray(keys[i])
            oracle.lib.ko_aes256ctr(ctypes.c_void_p(want.ctypes.data), want.size, ctypes.c_void_p(key.ctypes.data),
                                    ctypes.c_void_p(nonce.ctypes.data), first)
            assert np.array_equal(got[i].reshape(-1), want), (n, i)
