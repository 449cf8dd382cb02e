"""GPU: the canonical-output polynomial kernels (kyber_batch_poly_*_canon_dev, the wide fp32 engine
of kyber_amd/csrc/wpoly.h) against the oracle's restatement of the reference functions, canonicalised
as the reference's own serializers do (ref/poly.c:118-121): bit-exact in [0, q).

BASELINE.json config 2.  Sizes 1 / 8 / 13 / 1000 compare every polynomial with the oracle; 2^20
compares a strided sample with the oracle and every polynomial through size-independent
identities computed on the GPU."""
import numpy as np
import pytest
import torch

from kyber_amd import lib as kl

pytestmark = pytest.mark.gpu
Q = 3329
MONT = (1 << 16) % Q  # 2285


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _canon(a):
    return np.mod(np.asarray(a, np.int64), Q).astype(np.int16)


def _centred(a):
    """the representative in [-(q-1)/2, (q-1)/2]: what the reference's own callers feed invntt / basemul"""
    c = np.mod(np.asarray(a, np.int64), Q)
    return np.where(c > Q // 2, c - Q, c).astype(np.int16)


def _oracle_fused(oracle, a, b):
    """ref/poly.c:244-280 + ref/polyvec.c:200: invntt_tomont(poly_reduce(poly_ntt(a) o poly_ntt(b)))"""
    ta, tb = oracle.poly_op("ko_poly_ntt", a), oracle.poly_op("ko_poly_ntt", b)
    p = _centred(oracle.poly_basemul(ta, tb))
    return oracle.poly_op("ko_invntt_tomont", p)


def _check_all_ops(oracle, a, b):
    """a, b: int16[n, 256] the reference's functions can take (|.| <= 2q)"""
    n = a.shape[0]
    r = torch.empty((n, 256), dtype=torch.int16, device="cuda")
    got = kl.poly_ntt_canon_(_dev(a)).cpu().numpy()
    want_a = oracle.poly_op("ko_poly_ntt", a)
    assert np.array_equal(got, _canon(want_a)), "ntt_canon"
    assert got.min() >= 0 and got.max() < Q
    want_b = oracle.poly_op("ko_poly_ntt", b)
    # basemul on what poly_ntt leaves (centred) and on its canonical form: same residues
    want_p = _canon(oracle.poly_basemul(want_a, want_b))
    assert np.array_equal(kl.poly_basemul_canon(r, _dev(want_a), _dev(want_b)).cpu().numpy(), want_p), "basemul_canon"
    assert np.array_equal(kl.poly_basemul_canon(r, _dev(_canon(want_a)), _dev(_canon(want_b))).cpu().numpy(), want_p), "basemul_canon, canonical inputs"
    # invntt_tomont on reduced input, as ref/indcpa.c:286-287 / ref/polyvec.c:200 feed it, and on its canonical form
    red = _centred(want_p)
    want_i = _canon(oracle.poly_op("ko_invntt_tomont", red))
    assert np.array_equal(kl.poly_invntt_canon_(_dev(red)).cpu().numpy(), want_i), "invntt_canon"
    assert np.array_equal(kl.poly_invntt_canon_(_dev(want_p)).cpu().numpy(), want_i), "invntt_canon, canonical input"
    got_f = kl.poly_mul_fused_canon(r, _dev(a), _dev(b)).cpu().numpy()
    assert np.array_equal(got_f, _canon(_oracle_fused(oracle, a, b))), "mul_fused_canon"
    assert np.array_equal(got_f, want_i), "mul_fused_canon = invntt(basemul(ntt, ntt))"
    return got_f


@pytest.mark.parametrize("npoly", [1, 8, 13, 1000])
def test_canon_kernels_match_oracle(oracle, npoly):
    rng = np.random.default_rng(1000 + npoly)
    a = rng.integers(0, Q, (npoly, 256)).astype(np.int16)
    b = rng.integers(0, Q, (npoly, 256)).astype(np.int16)
    got = _check_all_ops(oracle, a, b)
    # algebra: basemul carries 2^-16, invntt_tomont 2^16 -> the plain product in Z_q[X]/(X^256+1)
    for i in range(min(npoly, 4)):
        full = np.convolve(a[i].astype(np.int64), b[i].astype(np.int64))
        neg = full[:256].copy()
        neg[:255] -= full[256:]
        assert np.array_equal(got[i], np.mod(neg, Q))


def test_canon_kernels_extremes_and_wide_inputs(oracle):
    """constant extremes, alternating signs, inputs up to 2q (the widest the reference's int16
    butterflies take), and every int16 value: beyond-range inputs are reduced mod q first"""
    rng = np.random.default_rng(5)
    a = rng.integers(-2 * Q, 2 * Q + 1, (64, 256)).astype(np.int16)
    b = rng.integers(-2 * Q, 2 * Q + 1, (64, 256)).astype(np.int16)
    a[0, :], b[0, :] = Q - 1, Q - 1
    a[1, :], b[1, :] = 0, Q - 1
    a[2, ::2], a[2, 1::2] = Q - 1, -(Q - 1)
    a[3, :], b[3, :] = 2 * Q, -2 * Q
    a[4, :], b[4, :] = (Q - 1) // 2, -((Q - 1) // 2)
    _check_all_ops(oracle, a, b)
    # the whole int16 range: equal to the result on the centred residues
    w = rng.integers(-32768, 32768, (32, 256)).astype(np.int16)
    w[0, :], w[1, :] = -32768, 32767
    v = rng.integers(-32768, 32768, (32, 256)).astype(np.int16)
    r = torch.empty((32, 256), dtype=torch.int16, device="cuda")
    wc, vc = _centred(w), _centred(v)
    assert np.array_equal(kl.poly_ntt_canon_(_dev(w)).cpu().numpy(), _canon(oracle.poly_op("ko_poly_ntt", wc)))
    assert np.array_equal(kl.poly_invntt_canon_(_dev(w)).cpu().numpy(), _canon(oracle.poly_op("ko_invntt_tomont", wc)))
    assert np.array_equal(kl.poly_basemul_canon(r, _dev(w), _dev(v)).cpu().numpy(), _canon(oracle.poly_basemul(wc, vc)))
    assert np.array_equal(kl.poly_mul_fused_canon(r, _dev(w), _dev(v)).cpu().numpy(), _canon(_oracle_fused(oracle, wc, vc)))


def test_canon_kernels_in_place(oracle):
    """r may be a or b (include/kyber_batch.h)"""
    rng = np.random.default_rng(6)
    a = rng.integers(0, Q, (100, 256)).astype(np.int16)
    b = rng.integers(0, Q, (100, 256)).astype(np.int16)
    r = torch.empty((100, 256), dtype=torch.int16, device="cuda")
    want_m = kl.poly_mul_fused_canon(r, _dev(a), _dev(b)).cpu().numpy()
    want_b = kl.poly_basemul_canon(r, _dev(a), _dev(b)).cpu().numpy()
    for fn, want in ((kl.poly_mul_fused_canon, want_m), (kl.poly_basemul_canon, want_b)):
        da, db = _dev(a), _dev(b)
        assert np.array_equal(fn(da, da, db).cpu().numpy(), want)
        da, db = _dev(a), _dev(b)
        assert np.array_equal(fn(db, da, db).cpu().numpy(), want)


def test_canon_kernels_full_size(oracle):
    """2^20 polynomials (BASELINE.json configs[1]): a strided sample against the oracle, every
    polynomial through identities evaluated on the GPU"""
    n = 1 << 20
    g = torch.Generator(device="cuda")
    g.manual_seed(0x4B59424552)
    a = torch.randint(0, Q, (n, 256), dtype=torch.int16, device="cuda", generator=g)
    b = torch.randint(0, Q, (n, 256), dtype=torch.int16, device="cuda", generator=g)
    ta, tb = kl.poly_ntt_canon_(a.clone()), kl.poly_ntt_canon_(b.clone())
    prod = kl.poly_basemul_canon(torch.empty_like(a), ta, tb)
    inv = kl.poly_invntt_canon_(prod.clone())
    fused = kl.poly_mul_fused_canon(torch.empty_like(a), a, b)
    for t in (ta, tb, prod, inv, fused):
        assert int(t.min()) >= 0 and int(t.max()) < Q
    # the fused kernel is the composition of the three
    assert torch.equal(fused, inv)
    # invntt_tomont(ntt(a)) = 2^16 a mod q
    back = kl.poly_invntt_canon_(ta.clone())
    assert torch.equal(back.to(torch.int32), (a.to(torch.int32) * MONT) % Q)
    # linearity: ntt(a + b) = ntt(a) + ntt(b) mod q
    s = kl.poly_ntt_canon_((a + b).contiguous())
    assert torch.equal(s.to(torch.int32), (ta.to(torch.int32) + tb.to(torch.int32)) % Q)
    # the Barrett-centred forward transform of the reference-exact API agrees mod q
    cen = kl.poly_ntt_(a.clone())
    assert int(cen.min()) >= -(Q // 2) and int(cen.max()) <= Q // 2
    assert torch.equal(cen.to(torch.int32) % Q, ta.to(torch.int32))
    idx = torch.cat([torch.arange(0, 64), torch.arange(64, n - 64, 257), torch.arange(n - 64, n)]).cuda()
    ha, hb = a[idx].cpu().numpy(), b[idx].cpu().numpy()
    want_a, want_b = oracle.poly_op("ko_poly_ntt", ha), oracle.poly_op("ko_poly_ntt", hb)
    assert np.array_equal(ta[idx].cpu().numpy(), _canon(want_a))
    assert np.array_equal(cen[idx].cpu().numpy(), want_a)
    assert np.array_equal(prod[idx].cpu().numpy(), _canon(oracle.poly_basemul(want_a, want_b)))
    assert np.array_equal(fused[idx].cpu().numpy(), _canon(_oracle_fused(oracle, ha, hb)))
