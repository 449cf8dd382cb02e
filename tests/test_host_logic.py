"""CPU: host-side logic and the C-ABI surface (no compute calls: no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

import kyber_amd
from kyber_amd import dist as kdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    """every function name include/*.h declares (macros expanded by hand)."""
    names = set()
    batch = open(os.path.join(ROOT, "include", "kyber_batch.h")).read()
    names.update(re.findall(r"^\s*(?:int|size_t|void|const char \*)\s*\*?(\w+)\(", batch, re.M))
    ml = open(os.path.join(ROOT, "include", "mlkem_batch.h")).read()
    names.update(re.findall(r"^\s*int\s+(\w+)\(", ml, re.M))
    for s in re.findall(r"^KYBER_HIP_DECLARE_SET\((\w+)\)", batch, re.M):
        names.update("%s_batch_%s" % (s, op) for op in ("keypair", "enc", "dec"))
    api = open(os.path.join(ROOT, "include", "kyber_api.h")).read()
    for s in re.findall(r"^KYBER_API_DECLARE\((\w+)\)", api, re.M):
        names.update("pqcrystals_%s_hip_%s" % (s, op) for op in ("keypair", "enc", "dec"))
    names.discard("NAME")
    return {n for n in names if not n.startswith("NAME") and not n.startswith("pqcrystals_##")}


def test_library_exports_every_declared_symbol():
    lib = kyber_amd.library()
    want = _declared_symbols()
    assert len(want) >= 18 + 18 + 20
    missing = [n for n in sorted(want) if not hasattr(lib, n)]
    assert not missing, missing


def test_sizes_match_reference_api_macros():
    # ref/api.h:6-9,29-32,52-55
    assert kyber_amd.sizes(2) == (800, 1632, 768)
    assert kyber_amd.sizes(3) == (1184, 2400, 1088)
    assert kyber_amd.sizes(4) == (1568, 3168, 1568)
    lib = kyber_amd.library()
    for k in (2, 3, 4):
        assert (lib.kyber_hip_pk_bytes(k), lib.kyber_hip_sk_bytes(k), lib.kyber_hip_ct_bytes(k)) == kyber_amd.sizes(k)
        assert all(s % 16 == 0 for s in kyber_amd.sizes(k))
    assert lib.kyber_hip_pk_bytes(5) == 0


def test_api_size_names_are_preprocessor_macros(tmp_path):
    """ref/api.h:11-14 spells the per-implementation sizes as macros; code that checks them with #if
    must compile against include/kyber_api.h too"""
    import subprocess
    lines = ['#include "kyber_api.h"']
    for name, base in (("kyber512", 2), ("kyber512_90s", 2), ("kyber768", 3), ("kyber768_90s", 3), ("kyber1024", 4), ("kyber1024_90s", 4)):
        pk, sk, ct = kyber_amd.sizes(base)
        for field, val in (("PUBLICKEYBYTES", pk), ("SECRETKEYBYTES", sk), ("CIPHERTEXTBYTES", ct), ("BYTES", 32)):
            lines += ["#if pqcrystals_%s_hip_%s != %d" % (name, field, val), "#error %s %s" % (name, field), "#endif"]
    src = tmp_path / "sizes.c"
    src.write_text("\n".join(lines) + "\nint main(void) { return 0; }\n")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "sizes.o")], check=True)


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = kyber_amd.library()
    buf = (ctypes.c_uint8 * 4096)()
    assert lib.kyber_batch_keypair(7, 0, 1, buf, buf, buf) == -1          # bad k
    assert lib.kyber_batch_enc(3, 0, 1, None, buf, buf, buf) == -1         # NULL output
    assert lib.kyber_batch_dec_dev(3, 0, 1, buf, None, buf, None) == -1
    # device pointers are 16-byte aligned (include/kyber_batch.h): refused before anything touches a GPU
    import ctypes as _ct
    base = _ct.addressof(buf)
    base += (-base) % 16
    al, mis = _ct.c_void_p(base), _ct.c_void_p(base + 8)
    assert lib.kyber_batch_dec_dev(3, 0, 1, al, mis, al, None) == -1
    assert b"16-byte aligned" in lib.kyber_hip_last_error()
    assert lib.kyber_batch_enc_dev(3, 0, 1, al, al, al, mis, None) == -1
    assert lib.kyber_batch_keypair_dev(3, 0, 1, mis, al, al, None) == -1
    assert lib.kyber_batch_dec_dev_noalloc(3, 0, 1, al, al, mis, None) == -1
    assert lib.kyber_batch_keypair(3, 0, 0, buf, buf, buf) == 0            # empty batch is a no-op
    assert lib.kyber_batch_enc(4, 0, 0, buf, buf, buf, None) == 0


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(kyber_amd.KyberHipError):
        kyber_amd.KyberHip(3).keypair(np.zeros((1, 64), np.uint8))


def test_division_by_q_multiplier_is_exact():
    """poly.h div_q: floor(x/3329) == (x * 20642679) >> 36 for every numerator the
    d-bit codecs can form ((x << 11) + 1664 with x < 3329, so x < 2^23)."""
    x = np.arange(1 << 23, dtype=np.uint64)
    assert np.array_equal((x * np.uint64(20642679)) >> np.uint64(36), x // np.uint64(3329))


def test_accept_step_trick():
    """kem_kernels.h accept_step: ((d - q) >> 30) & 2 == 2 iff d < q for 12-bit d."""
    d = np.arange(4096, dtype=np.uint32)
    step = ((d - np.uint32(3329)) >> np.uint32(30)) & np.uint32(2)
    assert np.array_equal(step == 2, d < 3329) and set(step.tolist()) == {0, 2}


def test_shard_bounds_partition():
    for n in (0, 1, 7, 8, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            cuts = [kdist.shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_float_compress_formula_is_exact():
    """fpoly.h fcompress<D>: low D mantissa bits of fl32(x * fl32(2^D/q) + 1.5*2^23)
    equal the reference's ((u << D) + q/2) / q & mask (ref/poly.c:30,45,187-190,
    ref/polyvec.c:26,50) for every integer x in [-5000, 5000], u = x mod q.
    The single-rounding FMA is reproduced in float64 (the sum is exact there)."""
    q = 3329
    x = np.arange(-5000, 5001, dtype=np.int64)
    u = x % q
    for d in (1, 4, 5, 10, 11):
        c = np.float32(np.float32(1 << d) / np.float32(q))
        y = (x.astype(np.float64) * np.float64(c) + 12582912.0).astype(np.float32)
        got = y.view(np.uint32) & ((1 << d) - 1)
        want = (((u << d) + q // 2) // q) & ((1 << d) - 1)
        assert np.array_equal(got, want), d
        # distance to the nearest rounding boundary really exceeds the constant's error
        frac = (u.astype(np.float64) * (1 << d) / q) % 1.0
        assert np.min(np.abs(frac[u != 0] - 0.5)) > 1.49e-4


def test_new_entry_points_reject_bad_arguments_without_a_gpu():
    """shared-key, key-exchange and tuning entry points: argument checks come
    before any HIP call (ref/kex.h and the batch API have no error returns;
    the negative codes are this library's)."""
    lib = kyber_amd.library()
    from kyber_amd import kex as kx
    kx._bind(lib)
    buf = (ctypes.c_uint8 * 8192)()
    assert lib.kyber_batch_enc_shared_pk(9, 0, 1, buf, buf, buf, buf) == -1
    assert lib.kyber_batch_dec_shared_sk(3, 0, 1, None, buf, buf) == -1
    assert lib.kyber_batch_enc_shared_pk(3, 1, 0, buf, buf, buf, None) == 0
    assert lib.kyber_batch_kex_uake_initA(3, 0, 1, buf, buf, buf, buf, buf, 2) == -1      # unknown flag
    assert lib.kyber_batch_kex_ake_sharedB(3, 0, 1, buf, buf, buf, buf, None, buf, 0) == -1  # NULL pka
    assert lib.kyber_batch_kex_ake_sharedA(5, 0, 1, buf, buf, buf, buf, buf) == -1
    assert lib.kyber_batch_kex_uake_sharedA(2, 1, 0, buf, buf, buf, buf) == 0
    assert lib.kyber_hip_set_chunk((1 << 24) + 1) == -1
    assert lib.kyber_hip_set_chunk(0) == 0
    assert lib.kyber_hip_set_tail_capacity((1 << 64) - 1) == 0
    assert lib.kyber_hip_set_workspace_slots(0) == -1 and lib.kyber_hip_set_workspace_slots(9) == -1
    assert lib.kyber_hip_set_workspace_slots(8) == 0 and lib.kyber_hip_set_workspace_slots(1) == 0


def test_16bit_montgomery_forms_match_reference_reduce(oracle):
    """field.h builds Montgomery products from v_mul_lo_u16 / v_mad_i32_i16:
    t = int16(low16(b) * low16(w_hi)); r = (z*b - t*q) >> 16 with the twiddle
    pair w = z | (z * q^-1 mod 2^16) << 16.  Same integers as
    montgomery_reduce(z*b) of ref/reduce.c:16-23 for every int16 b and every zeta."""
    q, qinv = 3329, 62209
    zetas = np.ctypeslib.as_array(oracle.lib.ko_zetas(), (128,)).astype(np.int64)
    b = np.arange(-32768, 32768, dtype=np.int64)
    for z in list(zetas[::9]) + [1441, -1044]:
        zq = (z * qinv) & 0xFFFF
        m = ((b & 0xFFFF) * zq) & 0xFFFF
        m = np.where(m >= 32768, m - 65536, m)          # the mad reads it as int16
        got = (z * b - m * q) >> 16
        a = z * b
        t = (a * qinv) & 0xFFFF
        t = np.where(t >= 32768, t - 65536, t)          # (int16_t)a * QINV  ref/reduce.c:20
        want = (a - t * q) >> 16
        assert np.array_equal(got, want)
        assert np.all((got * 65536 - a) % q == 0) and np.all(np.abs(got) < q)


def _fred32(x):
    """fpoly.h fred on integer-valued int64 arrays, the FMA reproduced in float64."""
    q, magic = 3329, 12582912.0
    k = (x.astype(np.float64) * np.float64(np.float32(1.0) / np.float32(q)) + magic).astype(np.float32).astype(np.float64) - magic
    return (x - (k * q).astype(np.int64))


def test_cached_multiplicand_accumulation_stays_exact():
    """fpoly.h FCache / f_corr / f_mac: base multiplication of canonical matrix
    entries a in [0, q) against a cached centred multiplicand as unreduced FMA
    chains that start at corr = -q * rne(1664 * S / q).  fp32 arithmetic on
    integers is exact while every intermediate stays below 2^24, so the model
    runs the kernel's operation order in int64 and checks (1) that bound for
    worst-case sign patterns and random data, for one group of three
    polynomials (Kyber768) and groups of two (Kyber512/1024), and (2) that the
    reduced result is congruent to the reference's sum of basemul terms
    (ref/ntt.c:139-146, ref/polyvec.c:194-198 without the Montgomery factors)."""
    q, lim = 3329, 1 << 24
    rng = np.random.default_rng(12)
    n = 20000
    for T in (2, 3):                       # polynomials per accumulation group
        for mode in ("random", "worst+", "worst-", "max"):
            b = rng.integers(-1665, 1666, (T, 2, n))      # b[j][0] = b[2p], b[j][1] = b[2p+1]
            zb = rng.integers(-1665, 1666, (T, n))         # (+-zeta) * b[2p+1] mod q, any centred value
            if mode == "max":
                b = np.where(rng.integers(0, 2, b.shape) == 1, 1665, -1665)
                zb = np.where(rng.integers(0, 2, zb.shape) == 1, 1665, -1665)
            a = rng.integers(0, q, (T, 2, n))
            if mode in ("worst+", "max"):  # a large exactly where the factor is positive: partial sums as positive as possible
                a[:, 0] = np.where(b[:, 0] > 0, q - 1, 0)
                a[:, 1] = np.where(zb > 0, q - 1, 0)
            if mode == "worst-":
                a[:, 0] = np.where(b[:, 0] > 0, 0, q - 1)
                a[:, 1] = np.where(zb > 0, 0, q - 1)
            for which in (0, 1):           # even output c[2p], odd output c[2p+1]
                f0 = b[:, 0] if which == 0 else b[:, 1]   # factor of a[2p]
                f1 = zb if which == 0 else b[:, 0]        # factor of a[2p+1]
                S = (f0 + f1).sum(axis=0)
                x = 1664 * S
                assert np.abs(x).max() < lim
                corr = -(x - _fred32(x))                  # -q * rne(x / q)
                assert np.all(corr % q == 0) and np.abs(corr + x).max() <= 1665
                acc = corr.copy()
                peak = np.abs(acc).max()
                for j in range(T):                        # f_mac: inner FMA first (a[2p+1] term), then the outer
                    acc = acc + a[j, 1] * f1[j]
                    peak = max(peak, np.abs(acc).max(), np.abs(a[j, 1] * f1[j]).max())
                    acc = acc + a[j, 0] * f0[j]
                    peak = max(peak, np.abs(acc).max())
                assert peak < lim, (T, mode, which, peak)
                want = sum(a[j, 0] * f0[j] + a[j, 1] * f1[j] for j in range(T))
                r = _fred32(acc)
                assert np.abs(r).max() <= 1665 and np.all((r - want) % q == 0)
    # key bytes: any 12-bit value, centred by fred before f_mac; six products of centred values
    a12 = np.arange(0, 4096, dtype=np.int64)
    ac = _fred32(a12)
    assert np.abs(ac).max() <= 1665 and np.all((ac - a12) % q == 0)
    assert 6 * 1665 * 1665 < lim
    # inverse transform without the final 1/128: outputs are sums of two reduced values,
    # and the compression input bound of test_float_compress_formula_is_exact still holds
    assert 2 * 1665 + 2 + 1665 <= 5000


def test_bench_work_model_adds_up():
    """bench.py prices every pipeline kernel per instance (kernel_work); over one encaps + decaps the
    Keccak kernels' lane-ops must add up to SURVEY §3.4's permutation counts (61 / 100 / 155), with
    and without a shared key pair, and the 90s model to lane_ops_per_encdec"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    P = bench.KECCAK_LANE_OPS
    for k, want in ((2, 61), (3, 100), (4, 155)):
        assert bench.perms_per_encdec(k) == want
        for shared in (False, True):
            w = lambda name: bench.kernel_work(name, k, 0, shared)
            enc = ["k_enc_pre", "k_prf", "k_post"] + ([] if shared else ["k_sample_matrix"])
            dec = ["k_dec_mid", "k_prf", "k_post"] + ([] if shared else ["k_sample_matrix"])
            total = sum(w(n)[0] for n in enc + dec)
            assert total == bench.perms_per_encdec(k, shared) * P, (k, shared)
            assert bench.lane_ops_per_encdec(k, 0, shared) == total
        # 90s: the symmetric kernels of an encaps + decaps are what lane_ops_per_encdec(k, 1) prices
        names = ["k90_enc_pre", "k90_sample_matrix", "k90_prf", "k90_post", "k90_dec_mid", "k90_sample_matrix", "k90_prf", "k90_post"]
        assert sum(bench.kernel_work(n, k, 1)[0] for n in names) == bench.lane_ops_per_encdec(k, 1)
    assert bench.kernel_work("k_sample_matrix_tail", 3, 0) is None and bench.kernel_work("k_sample_matrix", 3, 0, True) is None
    # compulsory bytes: an encapsulation reads pk + coins and writes ct + ss (2336 B for Kyber768)
    assert 1184 + 32 + 1088 + 32 == 2336


def test_wide_engine_products_and_reductions_are_exact():
    """kyber_amd/csrc/wpoly.h in numpy: fmulc (three FMAs), fred_canon and fmulc_canon (floor quotient)
    over their whole input ranges.  An fp32 FMA is modelled as float32(float64(a) * float64(b) + float64(c)):
    for these operand sizes the float64 sum is exact, or -- where it is not (fred_canon's quotient) -- never
    close enough to a rounding boundary for the second rounding to matter, which is asserted."""
    Q = 3329
    f32, f64 = np.float32, np.float64
    M, QINV = f32(12582912.0), f32(1.0) / f32(3329.0)
    QM = f32(3329.0) * M
    assert float(QM) == 3329 * 12582912  # q * magic is an fp32 number

    def fma(a, b, c):
        return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)

    def fmulc(b, z):
        zf = np.full_like(b, f32(z))
        zq = np.full_like(b, f32(z) * QINV)
        k = fma(b, zq, np.full_like(b, M))
        m = fma(k, np.full_like(b, f32(-Q)), np.full_like(b, QM))
        return fma(zf, b, m)

    brv7 = lambda i: int(format(i, "07b")[::-1], 2)
    cen = lambda v: (v % Q) - (Q if (v % Q) > Q // 2 else 0)
    consts = sorted({cen(pow(17, brv7(i), Q)) for i in range(128)} | {-cen(pow(17, brv7(i), Q)) for i in range(64, 128)}
                    | {169, 512, -266, -26, 1652})
    b = np.arange(-10078, 10079, dtype=np.int64)
    for z in consts:
        if abs(z) * 10078 >= 1 << 24:
            bb = b[np.abs(b) * abs(z) < (1 << 24)]
        else:
            bb = b
        r = fmulc(bb.astype(f32), z).astype(np.int64)
        assert np.array_equal(np.mod(r - z * bb, Q), np.zeros_like(bb)), z
        assert np.abs(r).max() <= 1665, (z, np.abs(r).max())

    def fred_canon(x):
        xs = (x.astype(f64) - 1664.0)
        exact = xs * f64(QINV) + f64(M)                       # may be rounded by float64 ...
        frac = np.abs(exact - np.floor(exact) - 0.5)
        assert frac.min() > 1e-6                              # ... but never near a half: both roundings agree
        k = exact.astype(f32)
        return x + fma(k, np.full_like(x, f32(-Q)), np.full_like(x, QM))

    xs = np.concatenate([np.arange(-(1 << 17), 1 << 17),
                         (np.arange(-5039, 5040)[:, None] * Q + np.array([0, 1, 2, 1663, 1664, 1665, 1666, 3327, 3328])).ravel()])
    xs = xs[np.abs(xs) < (1 << 24)]
    r = fred_canon(xs.astype(f32)).astype(np.int64)
    assert np.array_equal(r, np.mod(xs, Q))

    def fmulc_canon(b, z):
        zf = np.full_like(b, f32(z))
        p = fma(zf, b, np.full_like(b, f32(-1664.0)))
        exact = p.astype(f64) * f64(QINV) + f64(M)
        assert np.abs(exact - np.floor(exact) - 0.5).min() > 1e-6
        k = exact.astype(f32)
        return fma(zf, b, fma(k, np.full_like(b, f32(-Q)), np.full_like(b, QM)))

    bb = np.arange(-6660, 6661, dtype=np.int64)
    for z in (512, -266, -26, 1652, 1664, -1664):
        r = fmulc_canon(bb.astype(f32), z).astype(np.int64)
        assert np.array_equal(r, np.mod(z * bb, Q)), z


def test_pair_rotation_identity_of_keccak_split():
    """kyber_amd/csrc/keccak_split.h: with the low halves of the 25 lanes on one work-item and the high halves on
    its partner, a 64-bit rotate by N is own' = alignbit(own, other, 32 - N) for N < 32, `other` for N = 32 and
    alignbit(other, own, 64 - N) for N > 32 -- the same instruction on both work-items.  Checked here for every
    rho offset of Keccak-f[1600] and the theta rotate on random words (v_alignbit_b32: (hi:lo) >> s, low 32 bits)."""
    def alignbit(hi, lo, s):
        return ((hi << 32 | lo) >> s) & 0xFFFFFFFF

    def half(own, oth, n):
        if n == 0:
            return own
        if n == 32:
            return oth
        return alignbit(own, oth, 32 - n) if n < 32 else alignbit(oth, own, 64 - n)

    offsets, x, y = {0, 1}, 1, 0
    for t in range(24):
        offsets.add(((t + 1) * (t + 2) // 2) % 64)
        x, y = y, (2 * x + 3 * y) % 5
    assert len(offsets) == 25 and max(offsets) == 62
    rng = np.random.default_rng(64)
    for n in sorted(offsets):
        for v in [int(w) for w in rng.integers(0, 1 << 63, 50, dtype=np.uint64)] + [1, 1 << 63, (1 << 64) - 1]:
            lo, hi = v & 0xFFFFFFFF, v >> 32
            want = ((v << n) | (v >> (64 - n))) & ((1 << 64) - 1) if n else v
            assert half(lo, hi, n) == want & 0xFFFFFFFF and half(hi, lo, n) == want >> 32, n
