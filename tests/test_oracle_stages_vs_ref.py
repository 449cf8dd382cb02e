"""CPU: every stage function of the oracle against the reference's own function
of the same role, called directly in oracle/_ref/libkyber_ref_<set>.so (the
reference's ref/ sources compiled in place export their namespaced internals,
ref/params.h:11-31).  One test per row group of SURVEY.md 8(a): a2-a3
reductions, a4-a8 transforms and products, a9-a11 matrix expansion, a13 noise,
a14-a17 codecs, a12 Keccak front ends.  Skipped where oracle/_ref is absent."""
import ctypes

import numpy as np
import pytest

import oracle_lib
import vectors

Q = 3329


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _ref(k, n90):
    try:
        return oracle_lib.RefLib("ref", k, n90).lib, "pqcrystals_kyber%s_ref_" % oracle_lib.REF_SETS[(k, n90)]
    except (OSError, FileNotFoundError):
        pytest.skip("oracle/_ref not built")


def test_reductions_exhaustive(oracle):
    lib, pre = _ref(3, 0)
    bar = getattr(lib, pre + "barrett_reduce")
    bar.argtypes, bar.restype = [ctypes.c_int16], ctypes.c_int16
    mont = getattr(lib, pre + "montgomery_reduce")
    mont.argtypes, mont.restype = [ctypes.c_int32], ctypes.c_int16
    for a in range(-32768, 32768, 1):
        assert oracle.lib.ko_barrett_reduce(a) == bar(a), a
    rng = np.random.default_rng(1)
    for a in list(rng.integers(-Q * 32768 + 1, Q * 32768, 20000)) + [0, 1, -1, Q * 32768 - 1, -Q * 32768 + 1]:
        assert oracle.lib.ko_montgomery_reduce(int(a)) == mont(int(a)), a


@pytest.mark.parametrize("k,n90", [(2, 0), (3, 0), (4, 0)])
def test_transforms_and_products(oracle, k, n90):
    lib, pre = _ref(k, n90)
    rng = np.random.default_rng(k)
    n = 300
    a = rng.integers(-Q + 1, Q, (n, 256)).astype(np.int16)
    b = rng.integers(-Q + 1, Q, (n, 256)).astype(np.int16)
    for ours, theirs in (("ko_ntt", "ntt"), ("ko_poly_ntt", "poly_ntt"), ("ko_invntt_tomont", "invntt"), ("ko_poly_tomont", "poly_tomont")):
        want = a.copy()
        fn = getattr(lib, pre + theirs)
        for row in want:
            fn(_p(row))
        assert np.array_equal(oracle.poly_op(ours, a), want), theirs
    ta, tb = oracle.poly_op("ko_poly_ntt", a), oracle.poly_op("ko_poly_ntt", b)
    want = np.empty_like(ta)
    fn = getattr(lib, pre + "poly_basemul_montgomery")
    for i in range(n):
        fn(_p(want[i]), _p(ta[i]), _p(tb[i]))
    assert np.array_equal(oracle.poly_basemul(ta, tb), want)
    # polyvec_basemul_acc_montgomery on k-vectors
    fn = getattr(lib, pre + "polyvec_basemul_acc_montgomery")
    oracle.lib.ko_polyvec_basemul_acc.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    for i in range(0, n - k, k):
        va, vb = np.ascontiguousarray(ta[i:i + k]), np.ascontiguousarray(tb[i:i + k])
        w, g = np.empty(256, np.int16), np.empty(256, np.int16)
        fn(_p(w), _p(va), _p(vb))
        oracle.lib.ko_polyvec_basemul_acc(k, _p(g), _p(va), _p(vb))
        assert np.array_equal(g, w)


@pytest.mark.parametrize("k,n90", sorted(vectors.META_NAME))
def test_gen_matrix_and_noise(oracle, k, n90):
    lib, pre = _ref(k, n90)
    rng = np.random.default_rng(10 * k + n90)
    gm = getattr(lib, pre + "gen_matrix")
    for t in (0, 1):
        for _ in range(60):
            seed = rng.integers(0, 256, 32, dtype=np.uint8)
            want = np.empty((k * k, 256), np.int16)
            gm(_p(want), _p(seed), t)
            got, _ = oracle.gen_matrix(k, seed, t, n90)
            assert np.array_equal(got, want)
    p = oracle.params(k, n90)
    oracle.lib.ko_getnoise.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint8]
    for eta_name, eta in (("eta1", 3 if k == 2 else 2), ("eta2", 2)):
        fn = getattr(lib, pre + "poly_getnoise_" + eta_name)
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint8]
        for nonce in (0, 1, 5, 255):
            seed = rng.integers(0, 256, 32, dtype=np.uint8)
            w, g = np.empty(256, np.int16), np.empty(256, np.int16)
            fn(_p(w), _p(seed), nonce)
            oracle.lib.ko_getnoise(ctypes.byref(p), eta, _p(g), _p(seed), nonce)
            assert np.array_equal(g, w), (eta_name, nonce)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_codecs(oracle, k):
    lib, pre = _ref(k, 0)
    rng = np.random.default_rng(77 + k)
    dv, du = (5, 11) if k == 4 else (4, 10)
    for _ in range(100):
        a = rng.integers(-Q + 1, Q, 256).astype(np.int16)      # tobytes/compress accept (-q, q)
        w, g = np.empty(384, np.uint8), np.empty(384, np.uint8)
        getattr(lib, pre + "poly_tobytes")(_p(w), _p(a))
        oracle.lib.ko_poly_tobytes(_p(g), _p(a))
        assert np.array_equal(g, w)
        raw = rng.integers(0, 256, 384, dtype=np.uint8)
        pw, pg = np.empty(256, np.int16), np.empty(256, np.int16)
        getattr(lib, pre + "poly_frombytes")(_p(pw), _p(raw))
        oracle.lib.ko_poly_frombytes(_p(pg), _p(raw))
        assert np.array_equal(pg, pw)
        # poly_compress / poly_decompress (d_v)
        w, g = np.empty(32 * dv, np.uint8), np.empty(32 * dv, np.uint8)
        getattr(lib, pre + "poly_compress")(_p(w), _p(a))
        oracle.lib.ko_poly_compress(dv, _p(g), _p(a))
        assert np.array_equal(g, w)
        getattr(lib, pre + "poly_decompress")(_p(pw), _p(w))
        oracle.lib.ko_poly_decompress(dv, _p(pg), _p(w))
        assert np.array_equal(pg, pw)
        # polyvec_compress / polyvec_decompress (d_u), k polynomials
        va = rng.integers(-Q + 1, Q, (k, 256)).astype(np.int16)
        w = np.empty(32 * du * k, np.uint8)
        getattr(lib, pre + "polyvec_compress")(_p(w), _p(va))
        g = np.empty((k, 32 * du), np.uint8)
        for i in range(k):
            oracle.lib.ko_poly_compress(du, _p(g[i]), _p(va[i]))
        assert np.array_equal(g.reshape(-1), w)
        vw, vg = np.empty((k, 256), np.int16), np.empty((k, 256), np.int16)
        getattr(lib, pre + "polyvec_decompress")(_p(vw), _p(w))
        for i in range(k):
            oracle.lib.ko_poly_decompress(du, _p(vg[i]), _p(g[i]))
        assert np.array_equal(vg, vw)
        # message encoding
        msg = rng.integers(0, 256, 32, dtype=np.uint8)
        getattr(lib, pre + "poly_frommsg")(_p(pw), _p(msg))
        oracle.lib.ko_poly_frommsg(_p(pg), _p(msg))
        assert np.array_equal(pg, pw)
        mw, mg = np.empty(32, np.uint8), np.empty(32, np.uint8)
        getattr(lib, pre + "poly_tomsg")(_p(mw), _p(a))
        oracle.lib.ko_poly_tomsg(_p(mg), _p(a))
        assert np.array_equal(mg, mw)


def test_keccak_front_ends(oracle):
    lib, _ = _ref(3, 0)
    rng = np.random.default_rng(5)
    sz = ctypes.c_size_t
    for n in (0, 1, 33, 71, 72, 135, 136, 137, 167, 168, 169, 1184, 1568):
        data = rng.integers(0, 256, max(n, 1), dtype=np.uint8)[:n].copy()
        for ours, theirs, outlen in (("ko_sha3_256", "sha3_256", 32), ("ko_sha3_512", "sha3_512", 64)):
            w, g = np.empty(outlen, np.uint8), np.empty(outlen, np.uint8)
            fn = getattr(lib, "pqcrystals_kyber_fips202_ref_" + theirs)
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, sz]
            fn(_p(w), _p(data), n)
            getattr(oracle.lib, ours)(_p(g), _p(data), n)
            assert np.array_equal(g, w), (theirs, n)
        for ours, theirs in (("ko_shake128", "shake128"), ("ko_shake256", "shake256")):
            for outlen in (1, 32, 168, 169, 505):
                w, g = np.empty(outlen, np.uint8), np.empty(outlen, np.uint8)
                fn = getattr(lib, "pqcrystals_kyber_fips202_ref_" + theirs)
                fn.argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p, sz]
                fn(_p(w), outlen, _p(data), n)
                getattr(oracle.lib, ours)(_p(g), outlen, _p(data), n)
                assert np.array_equal(g, w), (theirs, n, outlen)
