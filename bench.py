#!/usr/bin/env python3
"""bench.py -- Kyber768 batched encaps+decaps throughput on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

One "step" = one pass of the hot path over one batch: B encapsulations
followed by B decapsulations (device-resident pk/sk/ct/ss, keys generated once
by the batched keypair before the timed region).  B defaults to 2^20 per GPU
(BASELINE.json configs[2]).  Instances are independent, so N GPUs run N
disjoint batches with no data-path collective ("weak" scaling); the only
torch.distributed traffic is the barrier and the max-over-ranks of the time.

--gpus N with N > 1 and no WORLD_SIZE in the environment: this process touches
no GPU and starts N rank processes of itself (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, rendezvous on 127.0.0.1), relays rank 0's JSON line and exits
with the worst child status.  Under torchrun (WORLD_SIZE set) it is one rank.
Either way the line is printed only if the world that ran equals --gpus.

Prints ONE JSON line (rank 0) with the whole-job rate, a `roofline` object for
the dominant kernel (k_sample_matrix: SHAKE128 matrix expansion + rejection
sampling) measured live with HIP events, an `ntt` object (BASELINE.json
configs[1], the second half of its metric), a `pcie_inclusive` object (the
host-pointer entry points, never the headline) and a `cpu_baseline` object:
the reference's own avx2/ and ref/ code from oracle/_ref timed on the host cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

K = 3  # Kyber768 (the headline workload); --workload switches the set for side measurements
N90 = 0
WORKLOADS = {  # name: (k, nineties, compulsory enc+dec bytes, Keccak-f per enc+dec)  SURVEY.md §3.4, §8(d)
    "kyber512": (2, 0, 4064, 61), "kyber768": (3, 0, 5856, 100), "kyber1024": (4, 0, 7968, 155),
    "kyber512-90s": (2, 1, 4064, 0), "kyber768-90s": (3, 1, 5856, 0), "kyber1024-90s": (4, 1, 7968, 0),
}
SEED0 = 0x4B59424552  # "KYBER", SURVEY.md §8(d)

# algorithmic constants (restated in DESIGN.md §4)
DEFAULT_CHUNK = 1 << 19            # kyber_hip.hip: instances per internal pass = units per launch
KECCAK_LANE_OPS = 6240           # 32-bit lane-ops per Keccak-f[1600] as written in the reference
VALU_PEAK_TLOPS = 78.6           # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, full-rate 32-bit int ops
HBM_PEAK_GBS = 8000.0
ENCDEC_BYTES = 5856              # compulsory HBM bytes per Kyber768 enc+dec
ENCDEC_PERMS = 100               # Keccak-f per Kyber768 enc+dec
MATRIX_ENTRY_PERMS = 3           # SHAKE128 blocks per matrix entry (4th block: 0.8 % of entries)
# 90s sets: 32-bit lane-ops of the textbook word-oriented forms
AES_BLOCK_LANE_OPS = 680         # AES-256 on one 16-byte block: 14 rounds x (16 look-ups + 16 byte selects + 16 xors) + whitening
SHA256_BLOCK_LANE_OPS = 2160     # 64 rounds x 24 + 48 schedule words x 13
SHA512_BLOCK_LANE_OPS = 5504     # 80 rounds x 24 x 2 + 64 schedule words x 13 x 2 (64-bit words on 32-bit lanes)
AES_MATRIX_ENTRY_BLOCKS = 32     # ref/indcpa.c:160-162,182: 8 x 64-byte XOF blocks before the first test
# polynomial kernels: lane-ops per transform
NTT_LANE_OPS, INVNTT_LANE_OPS, BASEMUL_LANE_OPS = 5400, 9000, 2800


def lane_ops_per_encdec(k, n90, shared=False):
    """Algorithmic 32-bit lane-ops of one encaps + one decaps (symmetric part only, as SURVEY §8(d) prices it)."""
    if not n90:
        return perms_per_encdec(k, shared) * KECCAK_LANE_OPS
    pkb, ctb = 384 * k + 32, 32 * ((11 if k == 4 else 10) * k + (5 if k == 4 else 4))
    sha256_blocks = lambda nbytes: (nbytes + 9 + 63) // 64
    prf_blocks = lambda eta: (64 * eta) // 16
    eta1 = 3 if k == 2 else 2
    enc_core = (0 if shared else k * k * AES_MATRIX_ENTRY_BLOCKS) + k * prf_blocks(eta1) + (k + 1) * prf_blocks(2)  # gen_matrix + noise
    aes = 2 * enc_core                                                   # encaps + decaps' re-encryption
    s256 = sha256_blocks(32) + (0 if shared else sha256_blocks(pkb)) + 2 * (sha256_blocks(ctb) + sha256_blocks(64))  # H(m0), H(pk), 2 x (H(ct), KDF)
    s512 = 2                                                             # G in encaps and decaps
    return aes * AES_BLOCK_LANE_OPS + s256 * SHA256_BLOCK_LANE_OPS + s512 * SHA512_BLOCK_LANE_OPS


def perms_per_encdec(k, shared=False):
    """Keccak-f per encaps + decaps.  shared: one key pair for the whole batch, so the
    matrix expansion and H(pk) are done once per call and drop out per instance."""
    pkb, ctb = 384 * k + 32, 32 * ((11 if k == 4 else 10) * k + (5 if k == 4 else 4))
    h = lambda nbytes: nbytes // 136 + 1
    prf = (2 if k == 2 else 1) * k + (k + 1)          # eta1 = 3 needs two blocks per polynomial (Kyber512)
    enc = 1 + 1 + prf + h(ctb) + 1                    # H(m0), G, noise, H(ct), KDF
    dec = 1 + prf + h(ctb) + 1                        # G, noise of the re-encryption, H(ct), KDF
    if not shared:
        enc += h(pkb) + 3 * k * k
        dec += 3 * k * k
    return enc + dec


def kernel_work(name, k, n90, shared=False):
    """Algorithmic work of ONE instance in one launch of a pipeline kernel: (32-bit lane-ops, compulsory HBM
    bytes), by the constants of SURVEY.md §8(d) / DESIGN.md §4, §4c.  None for kernels whose work is not
    per instance (the sampler's tail pass, the one-off hashes of the shared-key calls)."""
    pkb, skb = 384 * k + 32, 768 * k + 96
    ctb = 32 * ((11 if k == 4 else 10) * k + (5 if k == 4 else 4))
    eta1 = 3 if k == 2 else 2
    prf_bytes = lambda jobs_eta1, jobs_eta2: jobs_eta1 * 64 * eta1 + jobs_eta2 * 128
    h = lambda nbytes: nbytes // 136 + 1                                  # SHA3-256 permutations
    s256 = lambda nbytes: (nbytes + 9 + 63) // 64                          # SHA-256 compressions
    P, A, S2, S5 = KECCAK_LANE_OPS, AES_BLOCK_LANE_OPS, SHA256_BLOCK_LANE_OPS, SHA512_BLOCK_LANE_OPS
    mat_bytes = 0 if shared else k * k * 512
    t = {
        # SHAKE sets
        "k_sample_matrix": (k * k * MATRIX_ENTRY_PERMS * P, k * k * 512 + 32),
        "k_prf": (((2 if k == 2 else 1) * k + (k + 1)) * P, 32 + prf_bytes(k, k + 1)),
        "k_enc_pre": ((1 + (0 if shared else h(pkb)) + 1) * P, 32 + (0 if shared else pkb) + 96),
        "k_dec_mid": (1 * P, 64 + 64),
        "k_post": ((h(ctb) + 1) * P, ctb + 64 + 32),
        "k_kp_pre": (1 * P, 32 + 64),
        "k_kp_post": (h(pkb) * P, pkb + 64),
        # 90s sets
        "k90_sample_matrix": (k * k * AES_MATRIX_ENTRY_BLOCKS * A, k * k * 512 + 32),
        "k90_prf": ((k * 4 * eta1 + (k + 1) * 8) * A, 32 + prf_bytes(k, k + 1)),
        "k90_enc_pre": ((s256(32) + (0 if shared else s256(pkb))) * S2 + S5, 32 + (0 if shared else pkb) + 96),
        "k90_dec_mid": (S5, 64 + 64),
        "k90_post": ((s256(ctb) + s256(64)) * S2, ctb + 64 + 32),
        "k90_kp_pre": (S5, 32 + 64),
        "k90_kp_post": (s256(pkb) * S2, pkb + 64),
        # polynomial arithmetic (both families): NTT / base multiplications / inverse NTT per instance
        "k_enc_arith": (k * NTT_LANE_OPS + k * (k + 1) * BASEMUL_LANE_OPS + (k + 1) * INVNTT_LANE_OPS,
                        mat_bytes + prf_bytes(k, k + 1) + 384 * k + 32 + ctb),
        "k_dec_arith": (k * NTT_LANE_OPS + k * BASEMUL_LANE_OPS + INVNTT_LANE_OPS, ctb + 384 * k + 32),
        "k_kp_arith": (2 * k * NTT_LANE_OPS + k * k * BASEMUL_LANE_OPS, k * k * 512 + prf_bytes(2 * k, 0) + pkb + skb),
    }
    t["k_enc_arith_cmp"] = t["k_enc_arith"]
    # batches of up to 65536 instances (keccak_split.h): two independent stages in one launch
    add = lambda a, b: (t[a][0] + t[b][0], t[a][1] + t[b][1])
    t["k_enc_front"] = add("k_enc_pre", "k_sample_matrix")
    kp_prf = ((2 if k == 2 else 1) * 2 * k * P, 32 + prf_bytes(2 * k, 0))   # 2K noise polynomials, all eta1
    t["k_kp_front"] = (t["k_sample_matrix"][0] + kp_prf[0], t["k_sample_matrix"][1] + kp_prf[1])
    t["k_dec_middle"] = (t["k_sample_matrix"][0] + (h(ctb) + 1) * P, t["k_sample_matrix"][1] + ctb + 32 + 128)  # + H(ct) + G
    t["k_kdf_dec"] = (1 * P, 64 + 32 + 32)
    if shared and name in ("k_sample_matrix", "k90_sample_matrix"):  # one expansion per call, not per instance
        return None
    return t.get(name)


def lib_sha256():
    """identity of the in-tree libkyber_hip.so: the counter files record which build they measured"""
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "kyber_amd", "libkyber_hip.so"), "rb").read()).hexdigest()
    except OSError:
        return None


# pipeline kernel -> the exact instantiation that runs at 2^19 instances per launch, as rocprofv3 names it
def _pmc_kernel_name(kernel, k):
    par = "ky::Par<%d>" % k
    return {
        "k_sample_matrix": "ky::k_sample_matrix<%s, 512, 0>" % par, "k90_sample_matrix": "ky::k90_sample_matrix<%s, 1024, 0>" % par,
        "k_prf": "ky::k_prf<%s >" % par, "k90_prf": "ky::k90_prf<%s, 1024>" % par,
        "k_enc_arith": "ky::k_enc_arith<%s, false>" % par, "k_enc_arith_cmp": "ky::k_enc_arith<%s, true>" % par,
        "k_dec_arith": "ky::k_dec_arith<%s >" % par, "k_enc_pre": "ky::k_enc_pre<%s >" % par, "k_dec_mid": "ky::k_dec_mid<%s >" % par,
        "k90_enc_pre": "ky::k90_enc_pre<%s >" % par, "k90_dec_mid": "ky::k90_dec_mid<%s >" % par,
    }.get(kernel)


def _pmc_lookup(pattern, want):
    """newest profiles/<pattern> that holds kernel `want` -> (bytes per launch, file, stale?)"""
    import glob
    sha = lib_sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            d = json.load(open(path))
            v = d["kernels"].get(want)
        except (OSError, ValueError, KeyError):
            continue
        if v is not None:
            return v["hbm_bytes_per_launch"], os.path.relpath(path, ROOT), d.get("lib_sha256") != sha
    return None, None, None


def pmc_traffic(kernel, k, chunk, n90=0):
    """HBM bytes per launch of `kernel` from the newest committed PMC pass
    (profiles/rNN_pmc_traffic[_90s].json, written by tools/pmc_pass.sh +
    tools/pmc_summarize.py --traffic-json: separate FETCH_SIZE / WRITE_SIZE
    passes, FETCH_SIZE doubled as the MI355X guide prescribes for gfx950).
    Counters cannot be read from inside this process, so the figure is the
    recorded one for the same launch size and the same build of the library
    (the file carries the library's sha256), or None."""
    want = _pmc_kernel_name(kernel, k)
    if chunk != DEFAULT_CHUNK or want is None:
        return None, None
    v, src, stale = _pmc_lookup("r*_pmc_traffic_90s.json" if n90 else "r*_pmc_traffic.json", want)
    if v is None or stale:
        return None, (src + " (another build of the library: figure dropped)") if src else None
    return v, src


def pmc_traffic_ntt(batch, kernel="ky::k_wpoly<0, 2, 1>"):
    """The same for a polynomial kernel (profiles/rNN_pmc_traffic_ntt.json, recorded at the default
    batch of 2^20 polynomials per launch)."""
    if batch != 1 << 20:
        return None, None
    v, src, stale = _pmc_lookup("r*_pmc_traffic_ntt.json", kernel)
    if v is None or stale:
        return None, (src + " (another build of the library: figure dropped)") if src else None
    return v, src


def host_threads():
    """Threads for the CPU leg: the CPUs this process may run on, bounded by the
    box's per-job CPU share when the harness exports one (OMP_NUM_THREADS)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n)


def cpu_baseline(budget_s=20.0, sample=None):
    """The CPU leg (the only place this program touches oracle/): time the
    reference's own code (oracle/_ref, built from /root/reference's avx2/ and
    ref/ sources; falls back to the oracle port) on the host cores, one leg per
    implementation, and check a strided sample of what the GPU just produced
    against the same code.  sample = dict(kc, ec, pk, sk, ct, ss) of numpy arrays, or None."""
    import numpy as np
    import oracle_lib

    threads = host_threads()
    rng = np.random.default_rng(7)
    legs = []
    for cand in (("avx2",) if oracle_lib.has_avx2() else ()) + ("ref",):
        try:
            legs.append((cand, "reference", oracle_lib.RefLib(cand, K, N90)))
        except (OSError, FileNotFoundError):
            continue
    if not legs:
        o = oracle_lib.load_oracle()

        class _Port:
            def keypair(self, c): return o.keypair(K, c, N90)
            def enc(self, pk, c): return o.enc(K, pk, c, N90)
            def dec(self, ct, sk): return o.dec(K, ct, sk, N90)
        legs.append(("oracle", "port", _Port()))

    sample_ok = None
    if sample is not None:
        lib = legs[0][2]
        spk, ssk = lib.keypair(sample["kc"])
        sct, sss = lib.enc(spk, sample["ec"])
        sample_ok = bool(np.array_equal(spk, sample["pk"]) and np.array_equal(ssk, sample["sk"])
                         and np.array_equal(sct, sample["ct"]) and np.array_equal(sss, sample["ss"])
                         and np.array_equal(lib.dec(sct, ssk), sample["ss"]))
        assert sample_ok, "strided sample of the GPU results differs from the %s code" % legs[0][1]

    def time_leg(lib, budget):
        n0 = 256  # calibrate on one core
        kc, ec = rng.integers(0, 256, (n0, 64), dtype=np.uint8), rng.integers(0, 256, (n0, 32), dtype=np.uint8)
        pk, sk = lib.keypair(kc)
        t0 = time.perf_counter()
        ct, ss = lib.enc(pk, ec)
        ss2 = lib.dec(ct, sk)
        t1 = time.perf_counter() - t0
        assert (ss == ss2).all()
        one_core = n0 / t1
        per_thread = max(n0, int(one_core * budget * 0.5))
        kc = rng.integers(0, 256, (per_thread, 64), dtype=np.uint8)
        ec = rng.integers(0, 256, (per_thread, 32), dtype=np.uint8)
        pk, sk = lib.keypair(kc)

        def work():
            c, s = lib.enc(pk, ec)
            lib.dec(c, sk)

        ths = [threading.Thread(target=work) for _ in range(threads)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        return {"value": threads * per_thread / dt, "one_core": one_core,
                "sample": "enc+dec, %d instances per thread x %d threads (%.1f s)" % (per_thread, threads, dt)}

    out = None
    per_leg = budget_s / len(legs)
    for impl, kind, lib in legs:
        r = time_leg(lib, per_leg)
        r.update({"unit": "encaps+decaps/s", "cores": threads, "kind": kind, "impl": impl})
        if out is None:  # the first (fastest) implementation is the baseline object; the others ride along
            out = dict(r)
            out["host_cpus"] = os.cpu_count()
            out["gpu_sample_bit_exact"] = sample_ok
            out["legs"] = {}
        out["legs"][impl] = {"value": r["value"], "one_core": r["one_core"], "cores": threads, "kind": kind, "sample": r["sample"]}
    return out


# ---------------------------------------------------------------------------- config 2: polynomial kernels
def measure_ntt(n, steps, warmup):
    """BASELINE.json configs[1]: batched forward NTT / basemul / inverse NTT on
    n polynomials (int16[256]) resident in HBM.  -> dict per kernel with
    polys/s, GB/s and both roofline fractions."""
    import torch
    from kyber_amd import lib as kl
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(SEED0)
    a = torch.randint(0, 3329, (n, 256), dtype=torch.int16, device=dev, generator=g)
    b = torch.randint(0, 3329, (n, 256), dtype=torch.int16, device=dev, generator=g)
    r = torch.empty_like(a)
    res = {}

    todo = []

    def timed(name, fn, nbytes, lane_ops):
        todo.append((name, fn, nbytes, lane_ops))

    def run_all(rounds=3):
        """every kernel `steps` launches back to back per round, `rounds` rounds interleaved over the kernels
        (the clock a kernel sees depends on what ran before it); the median round is reported"""
        ms = {name: [] for name, *_ in todo}
        for name, fn, _, _ in todo:
            for _ in range(warmup):
                fn()
        for _ in range(rounds):
            for name, fn, _, _ in todo:
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()  # the kernels are launched on torch's current stream (kyber_amd.lib._torch_stream)
                for _ in range(steps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms[name].append(e0.elapsed_time(e1) / steps)
        for name, _, nbytes, lane_ops in todo:
            t = sorted(ms[name])[len(ms[name]) // 2]
            res[name] = {"ms": t, "polys_per_s": n / (t * 1e-3), "GBps": n * nbytes / (t * 1e-3) * 1e-9,
                         "hbm_frac": n * nbytes / (t * 1e-3) * 1e-9 / HBM_PEAK_GBS,
                         "valu_frac": n * lane_ops / (t * 1e-3) * 1e-12 / VALU_PEAK_TLOPS}

    work = a.clone()
    # the fast path: canonical outputs, wide fp32 engine (kyber_amd/csrc/wpoly.h)
    timed("poly_ntt", lambda: kl.poly_ntt_(work), 1024, NTT_LANE_OPS)  # Barrett-centred output, same engine
    timed("poly_ntt_canon", lambda: kl.poly_ntt_canon_(work), 1024, NTT_LANE_OPS)
    timed("poly_invntt_canon", lambda: kl.poly_invntt_canon_(work), 1024, INVNTT_LANE_OPS)
    timed("poly_basemul_canon", lambda: kl.poly_basemul_canon(r, a, b), 1536, BASEMUL_LANE_OPS)
    timed("poly_mul_fused_canon", lambda: kl.poly_mul_fused_canon(r, a, b), 1536, 2 * NTT_LANE_OPS + BASEMUL_LANE_OPS + INVNTT_LANE_OPS)
    # the reference-exact representatives (int16 Montgomery engine, the debugging API)
    work2 = a.clone()
    timed("poly_invntt", lambda: kl.poly_invntt_(work2), 1024, INVNTT_LANE_OPS)
    timed("poly_basemul", lambda: kl.poly_basemul(r, a, b), 1536, BASEMUL_LANE_OPS)
    timed("poly_mul_fused", lambda: kl.poly_mul_fused(r, a, b), 1536, 2 * NTT_LANE_OPS + BASEMUL_LANE_OPS + INVNTT_LANE_OPS)
    run_all()
    # the pipeline of configs[1] in one pass: ntt(a), ntt(b), basemul, invntt (four launches) vs the fused kernel
    wa, wb = a.clone(), b.clone()

    def chain():
        wa.copy_(a); wb.copy_(b)
        kl.poly_ntt_canon_(wa); kl.poly_ntt_canon_(wb)
        kl.poly_basemul_canon(r, wa, wb)
        kl.poly_invntt_canon_(r)
    chain()
    fused = kl.poly_mul_fused_canon(torch.empty_like(r), a, b)
    assert torch.equal(fused, r), "fused product differs from ntt -> basemul -> invntt"
    assert int(fused.min()) >= 0 and int(fused.max()) < 3329
    return res


def bench_ntt(args):
    import torch
    assert torch.cuda.is_available()
    res = measure_ntt(args.batch, args.steps, args.warmup)
    fwd = res["poly_ntt"]
    traffic, traffic_src = pmc_traffic_ntt(args.batch)
    print(json.dumps({
        "metric": "batched NTTs/sec per GPU", "value": fwd["polys_per_s"], "unit": "NTT/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": fwd["ms"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
        "config": {"workload": "forward NTT (poly_ntt) on %d polynomials of Z_3329[X]/(X^256+1), in place in HBM" % args.batch},
        "roofline": {"kernel": "k_wpoly<W_NTT>", "bound": "hbm", "achieved": fwd["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": fwd["hbm_frac"], "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes": args.batch * 1024.0},
        "kernels": res}), flush=True)


def measure_pcie(kem, n=1 << 17):
    """PCIe-inclusive rate: the host-pointer entry points on pinned host arrays
    (H2D of pk/coins, D2H of ct/ss, H2D of ct/sk, D2H of ss inside the timed
    region).  Never the headline value."""
    import ctypes
    import numpy as np
    from kyber_amd import lib as kl
    L = kl.library()
    rng = np.random.default_rng(1)
    kc, ec = rng.integers(0, 256, (n, 64), dtype=np.uint8), rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk0, sk0 = kem.keypair(kc)
    pk, sk, pec = kl.pinned_empty(pk0.shape), kl.pinned_empty(sk0.shape), kl.pinned_empty(ec.shape)
    pk[:], sk[:], pec[:] = pk0, sk0, ec
    ct, ss, ss2 = kl.pinned_empty((n, kem.ct_bytes)), kl.pinned_empty((n, 32)), kl.pinned_empty((n, 32))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        rc = L.kyber_batch_enc(kem.k, kem.nineties, n, p(ct), p(ss), p(pk), p(pec))
        rc |= L.kyber_batch_dec(kem.k, kem.nineties, n, p(ss2), p(ct), p(sk))
        dt = time.perf_counter() - t0
        assert rc == 0 and np.array_equal(ss, ss2)
        best = dt if best is None else min(best, dt)
    bytes_moved = n * (kem.pk_bytes + 32 + kem.ct_bytes + 32 + kem.ct_bytes + kem.sk_bytes + 32)
    return {"value": n / best, "unit": "encaps+decaps/s", "instances": n, "host_memory": "pinned",
            "link_GBps": bytes_moved / best * 1e-9,
            "note": "kyber_batch_enc + kyber_batch_dec on host pointers, copies inside the timed region; not the headline"}


# ---------------------------------------------------------------------------- launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """--gpus N outside torchrun: start N rank processes of this script.  The
    parent must not initialise the GPU (no torch import here)."""
    import tempfile
    port = _free_port()
    procs, errs = [], []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        errs.append(tempfile.TemporaryFile())  # every rank's stderr is kept: a rank that dies early says why
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[-1]))
    # Poll instead of blocking on rank 0: when one rank fails (bad device index, missing library) the
    # others would sit in the rendezvous until its time-out, so they are ended and the failure reported.
    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [i for i, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad and failed is None:
            failed = bad[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
        time.sleep(0.05)
    reader.join(timeout=10)
    rcs = [p.wait() for p in procs]
    if failed is None:
        failed = next((i for i, rc in enumerate(rcs) if rc != 0), None)
    out = b"".join(out_chunks)
    for r, fh in enumerate(errs):
        fh.seek(0)
        text = fh.read().decode(errors="replace")
        fh.close()
        if text.strip() and (r == 0 or rcs[r] != 0):
            sys.stderr.write("".join("[rank %d] %s\n" % (r, line) for line in text.splitlines()))
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with status %d; the other ranks were stopped\n" % (failed, rcs[failed]))
    for line in out.decode().splitlines():  # stdout carries the one JSON line; library chatter (e.g. "[Gloo] Rank 0 ...") goes to stderr
        (sys.stdout if line.startswith("{") and failed is None else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return abs(rcs[failed]) if failed is not None else 0


def load_engine(args):
    """The HIP engine.  KYBER_BENCH_ENGINE=module:factory (tests only; needs
    --device cpu) substitutes a stand-in so the N-rank plumbing can be rehearsed
    without a GPU; without it a missing library or GPU is an error."""
    spec = os.environ.get("KYBER_BENCH_ENGINE")
    if args.device == "cpu":
        if not spec:
            raise SystemExit("bench.py: --device cpu is a plumbing rehearsal and needs KYBER_BENCH_ENGINE; the product has no CPU path")
        mod, fn = spec.split(":")
        return getattr(__import__(mod), fn)(K, N90), None
    import kyber_amd
    from kyber_amd import lib as kl
    return kyber_amd.KyberHip(K, N90), kl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1 << 20, help="KEM instances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-measurements", action="store_true", help="skip the ntt and pcie_inclusive objects")
    ap.add_argument("--chunk", type=int, default=0, help="instances per internal pass (0 = library default)")
    ap.add_argument("--shared-key", action="store_true",
                    help="side measurement: every instance uses ONE key pair (kyber_batch_enc_shared_pk / dec_shared_sk)")
    ap.add_argument("--workload", default="kyber768", choices=sorted(WORKLOADS) + ["ntt"],
                    help="default kyber768 = BASELINE.json's metric; others are side measurements")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"], help="cpu: N-rank plumbing rehearsal with a stand-in engine (tests)")
    ap.add_argument("--dist-backend", default=None, choices=["nccl", "gloo"], help="default nccl (RCCL) on cuda, gloo on cpu")
    ap.add_argument("--ranks-share-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses device 0 (RCCL refuses two ranks on one GPU, so use --dist-backend gloo)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    if args.workload == "ntt":
        return bench_ntt(args)
    global K, N90, ENCDEC_BYTES, ENCDEC_PERMS
    K, N90, ENCDEC_BYTES, ENCDEC_PERMS = WORKLOADS[args.workload]
    set_name = "Kyber%d%s" % (256 * K, "-90s" if N90 else "")

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    on_gpu = args.device == "cuda"
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py needs a GPU"
        devidx = 0 if args.ranks_share_device else local
        torch.cuda.set_device(devidx)
        dev = torch.device("cuda", devidx)
    else:
        dev = torch.device("cpu")
    backend = args.dist_backend or ("nccl" if on_gpu else "gloo")
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torchrun, also with one rank, so the RCCL path can be rehearsed on one GPU
        import torch.distributed as dist
        kw = {"device_id": dev} if backend == "nccl" else {}
        # RCCL announces itself on STDOUT when its communicator comes up ("HIP version : ...", "Librccl path : ..."):
        # stdout carries the one JSON line and nothing else, so the file descriptor points at stderr until
        # the communicator exists (init + a first barrier).
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend, **kw)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    kem, kl = load_engine(args)
    if args.chunk:
        kl.set_chunk(args.chunk)
    n = args.batch
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED0 + rank)
    kcoins = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device=dev, generator=gen)
    ecoins = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=gen)
    pk, sk = kem.keypair_dev(kcoins[:1] if args.shared_key else kcoins)
    ct = torch.empty((n, kem.ct_bytes), dtype=torch.uint8, device=dev)
    ssb = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    ssa = torch.empty((n, 32), dtype=torch.uint8, device=dev)

    if args.shared_key:
        def step():
            kem.enc_shared_pk_dev(pk[0], ecoins, ct, ssb)
            kem.dec_shared_sk_dev(ct, sk[0], ssa)
    else:
        def step():
            kem.enc_dev(pk, ecoins, ct, ssb)
            kem.dec_dev(ct, sk, ssa)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # correctness of what was just timed: every decapsulated secret equals the
    # encapsulated one; a strided sample goes to the CPU leg for a bit-exact check
    assert torch.equal(ssa, ssb), "decaps != encaps shared secret"
    idx = torch.arange(0, n, 4096, device=dev)
    kidx = torch.zeros_like(idx) if args.shared_key else idx
    sample = {"kc": kcoins[kidx].cpu().numpy(), "ec": ecoins[idx].cpu().numpy(), "pk": pk[kidx].cpu().numpy(),
              "sk": sk[kidx].cpu().numpy(), "ct": ct[idx].cpu().numpy(), "ss": ssa[idx].cpu().numpy()}
    parity = None

    # per-kernel durations: a separate, untimed pass with HIP events on the launch stream
    roof = None
    kernels = {}
    if rank == 0 and kl is not None:
        kl.profile_enable(True)
        reps = 3
        for _ in range(reps):
            step()
        prof = kl.profile_read()
        kl.profile_enable(False)
        tot = sum(ms for _, ms in prof.values())
        chunk = min(n, args.chunk or DEFAULT_CHUNK)
        for name, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
            kernels[name] = {"launches": cnt, "avg_ms": ms / cnt, "share": ms / tot}
            w = kernel_work(name, K, N90, args.shared_key)
            if w is not None:  # one launch covers `chunk` instances
                avg_s = ms / cnt * 1e-3
                kernels[name].update({"valu_frac": chunk * w[0] / avg_s * 1e-12 / VALU_PEAK_TLOPS,
                                      "hbm_frac": chunk * w[1] / avg_s * 1e-9 / HBM_PEAK_GBS,
                                      "lane_ops_per_instance": w[0], "bytes_per_instance": w[1]})
        dom = max(((nm, v) for nm, v in prof.items() if kernel_work(nm, K, N90, args.shared_key)), key=lambda kv: kv[1][1])[0]
        cnt, ms = prof[dom]
        avg_s = ms / cnt * 1e-3
        ops1, bytes1 = kernel_work(dom, K, N90, args.shared_key)
        if dom in ("k_sample_matrix", "k90_sample_matrix"):
            units, unit_name = chunk * K * K, "matrix entries"
        else:
            units, unit_name = chunk, "instances"
        ops, bytes_ = chunk * ops1, float(chunk * bytes1)
        ach = ops / avg_s * 1e-12
        traffic, traffic_src = pmc_traffic(dom, K, chunk, N90)
        roof = {
            "kernel": dom, "bound": "valu", "achieved": ach, "peak": VALU_PEAK_TLOPS, "unit": "Tlane-op/s",
            "frac": ach / VALU_PEAK_TLOPS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes": bytes_, "avg_launch_ms": avg_s * 1e3,
            "units_per_launch": units, "unit_name": unit_name,
            "hbm": {"achieved": bytes_ / avg_s * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": bytes_ / avg_s * 1e-9 / HBM_PEAK_GBS},
        }

    if rank == 0:
        assert world == args.gpus
        total = n * world * args.steps
        value = total / dt
        out = {
            "metric": "%s encaps+decaps/sec%s" % (set_name, " (one shared key pair)" if args.shared_key else ""), "value": value, "unit": "encaps+decaps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("u32 (AES/SHA-2 words)" if N90 else "u32 (Keccak lanes as 2x32 bit)") + " + f32 (exact integer polynomial arithmetic)",
            "data": "synthetic (device-generated coins, seed 0x4B59424552)",
            "config": {"workload": "%s batched encaps+decaps%s, %d instances per GPU per step, device-resident"
                       % (set_name, " with one shared key pair" if args.shared_key else "", n),
                       "instances_per_gpu": n, "parallelism": "independent instances sharded per GPU, no collective",
                       "dist_backend": backend if dist is not None else None},
            "parity": {"ss_dec_equals_ss_enc": True, "strided_sample_vs_cpu_reference": parity},
            "pipeline": {"valu_frac": value / world * lane_ops_per_encdec(K, N90, args.shared_key) * 1e-12 / VALU_PEAK_TLOPS,
                         "hbm_frac": value / world * ENCDEC_BYTES * 1e-9 / HBM_PEAK_GBS,
                         "lane_ops_per_encdec": lane_ops_per_encdec(K, N90, args.shared_key)},
            "roofline": roof, "kernels": kernels,
        }
        if kl is None:
            out["engine"] = os.environ.get("KYBER_BENCH_ENGINE")
        if on_gpu and kl is not None and world == 1 and not args.no_side_measurements and not args.shared_key:
            try:
                out["ntt"] = measure_ntt(1 << 20, 10, 2)  # BASELINE.json configs[1]
                out["ntt"]["config"] = ("2^20 polynomials int16[256] in HBM; *_canon = canonical outputs (fast path); fused = ntt(a), ntt(b), basemul, "
                                        "invntt in one kernel, checked equal to the four-launch chain on all 2^20")
            except Exception as e:
                out["ntt"] = {"error": repr(e)}
            try:
                out["pcie_inclusive"] = measure_pcie(kem)
            except Exception as e:
                out["pcie_inclusive"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sample=sample)
                out["parity"]["strided_sample_vs_cpu_reference"] = out["cpu_baseline"]["gpu_sample_bit_exact"]
            except AssertionError:
                raise
            except Exception as e:  # the baseline must never sink the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
