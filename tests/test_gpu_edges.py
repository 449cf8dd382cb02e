"""GPU: edge inputs the reference accepts without validation (poly_frombytes
masks 12 bits and nothing else, ref/poly.c:142-143): keys, ciphertexts and
coins of all-0xFF / all-0x00 bytes put 4095 (or 0) in EVERY coefficient at
once -- the magnitude bound the fp32 engine's exactness rests on (fpoly.h) --
which random bytes never do.  Also batch sizes around the internal pass size.
Checked against the oracle and, where oracle/_ref is present, the reference's
own compiled code."""
import numpy as np
import pytest

import kyber_amd
from kyber_amd import lib as kl
import oracle_lib

pytestmark = pytest.mark.gpu

SETS = [(2, 0), (3, 0), (4, 0), (2, 1), (3, 1), (4, 1)]


def _checkers(oracle, k, n90):
    out = [("oracle", lambda c: oracle.keypair(k, c, n90), lambda pk, c: oracle.enc(k, pk, c, n90),
            lambda ct, sk: oracle.dec(k, ct, sk, n90))]
    for impl in ("ref", "avx2"):
        if impl == "avx2" and not oracle_lib.has_avx2():
            continue
        try:
            r = oracle_lib.RefLib(impl, k, n90)
        except (OSError, FileNotFoundError):
            continue
        out.append((impl, r.keypair, r.enc, r.dec))
    return out


def _fills(n, width, rng):
    """rows: all 0xFF, all 0x00, 0xFF/0x00 alternating by byte, 0xF0 pattern, then random"""
    a = rng.integers(0, 256, (n, width), dtype=np.uint8)
    a[0] = 0xFF
    a[1] = 0x00
    a[2, 0::2], a[2, 1::2] = 0xFF, 0x00
    a[3] = 0xF0
    a[4] = 0x0F
    return a


@pytest.mark.parametrize("k,n90", SETS)
def test_extreme_keys_ciphertexts_and_coins(oracle, k, n90):
    kem = kyber_amd.KyberHip(k, n90)
    rng = np.random.default_rng(900 + 10 * k + n90)
    n = 16
    kc, ec = _fills(n, 64, rng), _fills(n, 32, rng)
    pk, sk = kem.keypair(kc)                       # honest keys from extreme coins
    xpk = _fills(n, kem.pk_bytes, rng)             # extreme "public keys": t^ = 4095 everywhere, rho = 0xFF..
    xpk[5, :384 * k] = 0xFF                        # 4095 everywhere with a random rho
    xpk[6] = pk[6]
    xpk[6, :384 * k] = 0xFF
    ct, ss = kem.enc(pk, ec)
    xct, xss = kem.enc(xpk, ec)
    xsk = _fills(n, kem.sk_bytes, rng)             # extreme secret keys (s^, embedded pk, H, z all 0xFF / 0)
    xsk[5] = sk[5]
    xsk[5, :384 * k] = 0xFF                        # s^ = 4095 everywhere, rest honest
    yct = _fills(n, kem.ct_bytes, rng)             # extreme ciphertexts
    d1 = kem.dec(ct, sk)                           # honest
    d2 = kem.dec(yct, sk)                          # extreme ct, honest sk  -> rejection path
    d3 = kem.dec(yct, xsk)                         # extreme both
    d4 = kem.dec(xct, xsk)
    assert np.array_equal(d1, ss)
    for name, kp, enc, dec in _checkers(oracle, k, n90):
        opk, osk = kp(kc)
        assert np.array_equal(pk, opk) and np.array_equal(sk, osk), name
        oct_, oss = enc(pk, ec)
        assert np.array_equal(ct, oct_) and np.array_equal(ss, oss), name
        oxct, oxss = enc(xpk, ec)
        assert np.array_equal(xct, oxct), "%s: ciphertext for extreme pk" % name
        assert np.array_equal(xss, oxss), name
        assert np.array_equal(d2, dec(yct, sk)), name
        assert np.array_equal(d3, dec(yct, xsk)), name
        assert np.array_equal(d4, dec(xct, xsk)), name


@pytest.mark.parametrize("k,n90", [(3, 0), (2, 0), (4, 1)])
def test_batch_sizes_around_the_internal_pass(oracle, k, n90):
    """kyber_hip_set_chunk(c): n = c-1, c, c+1 and 2c+1 cross the pass boundary
    (ragged last pass of one instance); same bytes as one pass."""
    kem = kyber_amd.KyberHip(k, n90)
    c = 256
    rng = np.random.default_rng(4242 + k)
    nmax = 2 * c + 1
    kc = rng.integers(0, 256, (nmax, 64), dtype=np.uint8)
    ec = rng.integers(0, 256, (nmax, 32), dtype=np.uint8)
    opk, osk = oracle.keypair(k, kc, n90)
    oct_, oss = oracle.enc(k, opk, ec, n90)
    try:
        kl.set_chunk(c)
        for n in (c - 1, c, c + 1, nmax):
            pk, sk = kem.keypair(kc[:n])
            ct, ss = kem.enc(pk, ec[:n])
            ss2 = kem.dec(ct, sk)
            assert np.array_equal(pk, opk[:n]) and np.array_equal(sk, osk[:n]), n
            assert np.array_equal(ct, oct_[:n]) and np.array_equal(ss, oss[:n]) and np.array_equal(ss2, oss[:n]), n
    finally:
        kl.set_chunk(0)


@pytest.mark.parametrize("k,n90", [(4, 0), (3, 1)])
def test_one_million_instances_other_sets(oracle, k, n90):
    """One GPU's shard of BASELINE.json configs[3] (Kyber1024, 2^20 per GPU) and
    configs[4] (Kyber768-90s, 2^20): round trip for every instance, strided
    sample bit-exact against the oracle, rejection of a corrupted subset."""
    import torch
    n = 1 << 20
    kem = kyber_amd.KyberHip(k, n90)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x4B59424552 + k + n90)
    kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    pk, sk = kem.keypair_dev(kc)
    ct, ssb = kem.enc_dev(pk, ec)
    ssa = kem.dec_dev(ct, sk)
    assert torch.equal(ssa, ssb)
    assert torch.equal(sk[:, 384 * k:384 * k + kem.pk_bytes], pk)
    idx = torch.arange(0, n, 8191, device="cuda")
    opk, osk = oracle.keypair(k, kc[idx].cpu().numpy(), n90)
    oct_, oss = oracle.enc(k, opk, ec[idx].cpu().numpy(), n90)
    assert np.array_equal(pk[idx].cpu().numpy(), opk) and np.array_equal(sk[idx].cpu().numpy(), osk)
    assert np.array_equal(ct[idx].cpu().numpy(), oct_) and np.array_equal(ssb[idx].cpu().numpy(), oss)
    ct[::16, 100] ^= 0x80
    ssr = kem.dec_dev(ct, sk)
    same = (ssr == ssb).all(dim=1)
    assert not same[::16].any()
    mask = torch.ones(n, dtype=torch.bool, device="cuda")
    mask[::16] = False
    assert same[mask].all()
    j = torch.arange(0, n, 16 * 509, device="cuda")
    assert np.array_equal(ssr[j].cpu().numpy(), oracle.dec(k, ct[j].cpu().numpy(), sk[j].cpu().numpy(), n90))
    del pk, sk, ct, ssa, ssb, ssr, kc, ec
    torch.cuda.empty_cache()


@pytest.mark.parametrize("k", [2, 3, 4])
def test_batch_sizes_around_the_workgroup_switch(oracle, k):
    """The sponge kernels run one-wavefront workgroups below 131072 sponges per launch and
    512-lane phase-locked workgroups from there on (keccak.h sponge_wg): batch sizes that put
    the matrix sampler (K*K sponges per instance), the PRF (2K+1) and the hashes (1) on either
    side of the switch, last workgroup ragged.  Strided sample against the oracle, every
    instance round-trips."""
    import torch
    kem = kyber_amd.KyberHip(k, 0)
    g = torch.Generator(device="cuda")
    g.manual_seed(77 + k)
    switch = 131072
    sizes = sorted({switch // (k * k), switch // (k * k) + 1, switch // (2 * k + 1) + 1, switch - 1, switch + 1})
    for n in sizes:
        kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
        ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
        pk, sk = kem.keypair_dev(kc)
        ct, ssb = kem.enc_dev(pk, ec)
        ssa = kem.dec_dev(ct, sk)
        assert torch.equal(ssa, ssb), n
        idx = torch.cat([torch.arange(0, n, 997, device="cuda"), torch.arange(n - 70, n, device="cuda")])
        opk, osk = oracle.keypair(k, kc[idx].cpu().numpy(), 0)
        oct_, oss = oracle.enc(k, opk, ec[idx].cpu().numpy(), 0)
        assert np.array_equal(pk[idx].cpu().numpy(), opk) and np.array_equal(sk[idx].cpu().numpy(), osk), n
        assert np.array_equal(ct[idx].cpu().numpy(), oct_) and np.array_equal(ssb[idx].cpu().numpy(), oss), n
        del pk, sk, ct, ssa, ssb, kc, ec
    torch.cuda.empty_cache()


@pytest.mark.parametrize("k", [2, 3, 4])
def test_90s_batch_sizes_around_the_aes_workgroup_switches(oracle, k):
    """The AES kernels of the 90s sets run 256-, 512- or 1024-lane workgroups by the number of
    jobs per launch against the CU count (aes.h aes_wg: up to 256 and 512 jobs per CU): batch sizes
    that put the matrix sampler (K*K jobs per instance) and the PRF (2K+1) on either side of both
    switches, last workgroup ragged.  Strided sample against the oracle, every instance round-trips."""
    import torch
    kem = kyber_amd.KyberHip(k, 1)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    g = torch.Generator(device="cuda")
    g.manual_seed(190 + k)
    sizes = set()
    for per_cu in (256, 512):
        for jobs in (k * k, 2 * k + 1):
            sizes |= {cus * per_cu // jobs, cus * per_cu // jobs + 1}
    for n in sorted(sizes):
        kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
        ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
        pk, sk = kem.keypair_dev(kc)
        ct, ssb = kem.enc_dev(pk, ec)
        ssa = kem.dec_dev(ct, sk)
        assert torch.equal(ssa, ssb), n
        idx = torch.cat([torch.arange(0, n, 997, device="cuda"), torch.arange(n - 70, n, device="cuda")])
        opk, osk = oracle.keypair(k, kc[idx].cpu().numpy(), 1)
        oct_, oss = oracle.enc(k, opk, ec[idx].cpu().numpy(), 1)
        assert np.array_equal(pk[idx].cpu().numpy(), opk) and np.array_equal(sk[idx].cpu().numpy(), osk), n
        assert np.array_equal(ct[idx].cpu().numpy(), oct_) and np.array_equal(ssb[idx].cpu().numpy(), oss), n
        del pk, sk, ct, ssa, ssb, kc, ec
    torch.cuda.empty_cache()
