"""GPU against THE REFERENCE ITSELF: oracle/_ref holds the reference's own ref/
and avx2/ sources compiled in the build container (oracle/Makefile); the built
objects travel to the GPU box, the sources do not.  Tens of thousands of random
instances per parameter set, including corrupted ciphertexts (implicit
rejection, ref/kem.c:116-125).  Skipped where oracle/_ref is absent."""
import threading

import numpy as np
import pytest

import kyber_amd
import oracle_lib
import vectors

pytestmark = pytest.mark.gpu


def _ref(k, n90):
    for impl in (("avx2",) if oracle_lib.has_avx2() else ()) + ("ref",):
        try:
            return oracle_lib.RefLib(impl, k, n90), impl
        except (OSError, FileNotFoundError):
            continue
    pytest.skip("oracle/_ref not built")


def _threaded(fn, n, parts=8):
    """run fn(lo, hi) over [0, n) in slices on `parts` host threads (the shim releases the GIL in C)"""
    bounds = np.linspace(0, n, parts + 1).astype(int)
    ths = [threading.Thread(target=fn, args=(bounds[i], bounds[i + 1])) for i in range(parts)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()


@pytest.mark.parametrize("k,n90", sorted(vectors.META_NAME))
def test_gpu_equals_reference_on_many_random_instances(k, n90):
    ref, impl = _ref(k, n90)
    n = 100000 if impl == "avx2" else 6000
    rng = np.random.default_rng(31337 + 10 * k + n90)
    kc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    kem = kyber_amd.KyberHip(k, n90)
    pk, sk = kem.keypair(kc)
    ct, ss = kem.enc(pk, ec)
    bad = ct.copy()
    pos = rng.integers(0, ct.shape[1], size=n)
    bad[np.arange(n), pos] ^= rng.integers(1, 256, size=n, dtype=np.uint8)
    bad[::2] = ct[::2]
    got = kem.dec(bad, sk)

    rpk, rsk = np.empty_like(pk), np.empty_like(sk)
    rct, rss, rdec = np.empty_like(ct), np.empty_like(ss), np.empty_like(ss)

    def work(lo, hi):
        rpk[lo:hi], rsk[lo:hi] = ref.keypair(kc[lo:hi])
        rct[lo:hi], rss[lo:hi] = ref.enc(rpk[lo:hi], ec[lo:hi])
        rdec[lo:hi] = ref.dec(bad[lo:hi], rsk[lo:hi])

    _threaded(work, n)
    assert np.array_equal(pk, rpk), "pk"
    assert np.array_equal(sk, rsk), "sk"
    assert np.array_equal(ct, rct), "ct"
    assert np.array_equal(ss, rss), "ss (enc)"
    assert np.array_equal(got, rdec), "ss (dec, half of them rejected)"
    assert np.array_equal(got[::2], ss[::2]) and not np.any(np.all(got[1::2] == ss[1::2], axis=1))
