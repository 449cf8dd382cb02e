"""CPU: the oracle against the reference itself (oracle/_ref: the reference's
ref/ and avx2/ sources compiled in place) on random coins, all six sets.
Skipped where oracle/_ref has not been built (it cannot be built on the GPU box)."""
import numpy as np
import pytest

import oracle_lib
import vectors


def _lib(impl, k, n90):
    if impl == "avx2" and not oracle_lib.has_avx2():
        pytest.skip("host CPU lacks avx2/bmi2/aes")
    try:
        return oracle_lib.RefLib(impl, k, n90)
    except (OSError, FileNotFoundError):
        pytest.skip("oracle/_ref not built")


@pytest.mark.parametrize("impl", ["ref", "avx2"])
@pytest.mark.parametrize("k,n90", sorted(vectors.META_NAME))
def test_oracle_equals_reference(oracle, impl, k, n90):
    ref = _lib(impl, k, n90)
    n = 200
    rng = np.random.default_rng(1000 * k + n90)
    kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    rpk, rsk = ref.keypair(kc)
    opk, osk = oracle.keypair(k, kc, n90)
    assert np.array_equal(rpk, opk) and np.array_equal(rsk, osk)
    rct, rss = ref.enc(rpk, ec)
    oct_, oss = oracle.enc(k, opk, ec, n90)
    assert np.array_equal(rct, oct_) and np.array_equal(rss, oss)
    bad = rct.copy()
    bad[::2, 5] ^= 0x40  # half the batch takes the implicit-rejection path
    assert np.array_equal(ref.dec(bad, rsk), oracle.dec(k, bad, osk, n90))
    assert np.array_equal(oracle.dec(k, oct_, osk, n90), oss)
