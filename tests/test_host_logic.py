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
    for s in re.findall(r"^KYBER_HIP_DECLARE_SET\((\w+)\)", batch, re.M):
        names.update("%s_batch_%s" % (s, op) for op in ("keypair", "enc", "dec"))
    api = open(os.path.join(ROOT, "include", "kyber_api.h")).read()
    for s in re.findall(r"^KYBER_API_DECLARE\((\w+),", api, re.M):
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


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = kyber_amd.library()
    buf = (ctypes.c_uint8 * 4096)()
    assert lib.kyber_batch_keypair(7, 0, 1, buf, buf, buf) == -1          # bad k
    assert lib.kyber_batch_enc(3, 0, 1, None, buf, buf, buf) == -1         # NULL output
    assert lib.kyber_batch_dec_dev(3, 0, 1, buf, None, buf, None) == -1
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
