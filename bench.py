#!/usr/bin/env python3
"""bench.py -- Kyber768 batched encaps+decaps throughput on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

One "step" = one pass of the hot path over one batch: B encapsulations
followed by B decapsulations (device-resident pk/sk/ct/ss, keys generated once
by the batched keypair before the timed region).  B defaults to 2^20 per GPU
(BASELINE.json configs[2]).  Instances are independent, so N GPUs run N
disjoint batches with no data-path collective ("weak" scaling); the only
torch.distributed traffic is the barrier and the max-over-ranks of the time.

Prints ONE JSON line (rank 0) with the whole-job rate, a `roofline` object for
the dominant kernel (k_sample_matrix: SHAKE128 matrix expansion + rejection
sampling) measured live with HIP events, and a `cpu_baseline` object: the
reference's own avx2/ (or ref/) code from oracle/_ref timed on the host cores.
"""
import argparse
import json
import os
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

# algorithmic constants (restated in DESIGN.md)
DEFAULT_CHUNK = 1 << 19            # kyber_hip.hip: instances per internal pass = units per launch
KECCAK_LANE_OPS = 6240           # 32-bit lane-ops per Keccak-f[1600] as written in the reference
VALU_PEAK_TLOPS = 78.6           # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, full-rate 32-bit int ops
HBM_PEAK_GBS = 8000.0
ENCDEC_BYTES = 5856              # compulsory HBM bytes per Kyber768 enc+dec
ENCDEC_PERMS = 100               # Keccak-f per Kyber768 enc+dec
MATRIX_ENTRY_PERMS = 3           # SHAKE128 blocks per matrix entry (4th block: 0.8 % of entries)
MATRIX_ENTRY_BYTES = 512 + 32 / 9.0


def pmc_traffic(kernel, k, chunk):
    """HBM bytes per launch of `kernel` from the newest committed PMC pass
    (profiles/rNN_pmc_traffic.json, written by tools/pmc_pass.sh +
    tools/pmc_summarize.py --traffic-json: separate FETCH_SIZE / WRITE_SIZE
    passes, FETCH_SIZE doubled as the MI355X guide prescribes for gfx950).
    Counters cannot be read from inside this process, so the figure is the
    recorded one for the same launch size, or None."""
    import glob
    if chunk != DEFAULT_CHUNK:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            ks = json.load(open(path))["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        for name, v in ks.items():
            if name.startswith("ky::%s<ky::Par<%d>" % (kernel, k)):
                return v["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline(budget_s=20.0, sample=None):
    """The CPU leg (the only place this program touches oracle/): time the
    reference's own code (oracle/_ref, built from /root/reference's ref/ and
    avx2/ sources; falls back to the oracle port) on the host cores, and check
    a strided sample of what the GPU just produced against the same code.
    sample = dict(kc, ec, pk, sk, ct, ss) of numpy arrays, or None."""
    import numpy as np
    import oracle_lib

    threads = min(16, os.cpu_count() or 1)
    rng = np.random.default_rng(7)
    kind, impl, lib = "port", "oracle", None
    for cand in (("avx2",) if oracle_lib.has_avx2() else ()) + ("ref",):
        try:
            lib = oracle_lib.RefLib(cand, K, N90)
            kind, impl = "reference", cand
            break
        except (OSError, FileNotFoundError):
            continue
    if lib is None:
        o = oracle_lib.load_oracle()

        class _Port:
            def keypair(self, c): return o.keypair(K, c, N90)
            def enc(self, pk, c): return o.enc(K, pk, c, N90)
            def dec(self, ct, sk): return o.dec(K, ct, sk, N90)
        lib = _Port()

    sample_ok = None
    if sample is not None:
        spk, ssk = lib.keypair(sample["kc"])
        sct, sss = lib.enc(spk, sample["ec"])
        sample_ok = bool(np.array_equal(spk, sample["pk"]) and np.array_equal(ssk, sample["sk"])
                         and np.array_equal(sct, sample["ct"]) and np.array_equal(sss, sample["ss"])
                         and np.array_equal(lib.dec(sct, ssk), sample["ss"]))
        assert sample_ok, "strided sample of the GPU results differs from the %s code" % kind

    # calibrate on one core
    n0 = 512
    kc, ec = rng.integers(0, 256, (n0, 64), dtype=np.uint8), rng.integers(0, 256, (n0, 32), dtype=np.uint8)
    pk, sk = lib.keypair(kc)
    t0 = time.perf_counter()
    ct, ss = lib.enc(pk, ec)
    ss2 = lib.dec(ct, sk)
    t1 = time.perf_counter() - t0
    assert (ss == ss2).all()
    one_core = n0 / t1
    per_thread = max(n0, int(one_core * budget_s * 0.5))
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
    return {
        "value": threads * per_thread / dt, "unit": "encaps+decaps/s", "cores": threads, "kind": kind,
        "impl": impl, "one_core": one_core,
        "sample": "enc+dec, %d instances per thread x %d threads (%.1f s)" % (per_thread, threads, dt),
        "gpu_sample_bit_exact": sample_ok,
    }


def bench_ntt(args):
    """BASELINE.json configs[1]: batched forward NTT / basemul / inverse NTT on
    2^20 polynomials (int16[256]) resident in HBM; one JSON line."""
    import torch
    from kyber_amd import lib as kl
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    n = args.batch
    g = torch.Generator(device=dev)
    g.manual_seed(SEED0)
    a = torch.randint(0, 3329, (n, 256), dtype=torch.int16, device=dev, generator=g)
    b = torch.randint(0, 3329, (n, 256), dtype=torch.int16, device=dev, generator=g)
    r = torch.empty_like(a)
    res = {}

    def timed(name, fn, nbytes):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        res[name] = {"ms": ms, "polys_per_s": n / (ms * 1e-3), "GBps": n * nbytes / (ms * 1e-3) * 1e-9,
                     "hbm_frac": n * nbytes / (ms * 1e-3) * 1e-9 / HBM_PEAK_GBS}

    work = a.clone()
    timed("poly_ntt", lambda: kl.poly_ntt_(work), 1024)
    timed("poly_invntt", lambda: kl.poly_invntt_(work), 1024)
    timed("poly_basemul", lambda: kl.poly_basemul(r, a, b), 1536)
    timed("poly_mul_fused(ntt,ntt,basemul,invntt)", lambda: kl.poly_mul_fused(r, a, b), 1536)
    fwd = res["poly_ntt"]
    print(json.dumps({
        "metric": "batched NTTs/sec per GPU", "value": fwd["polys_per_s"], "unit": "NTT/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": fwd["ms"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
        "config": {"workload": "forward NTT (poly_ntt) on %d polynomials of Z_3329[X]/(X^256+1), in place in HBM" % n},
        "roofline": {"kernel": "k_poly_ntt", "bound": "hbm", "achieved": fwd["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": fwd["hbm_frac"], "traffic": None},
        "kernels": res}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1 << 20, help="KEM instances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=0, help="instances per internal pass (0 = library default)")
    ap.add_argument("--shared-key", action="store_true",
                    help="side measurement: every instance uses ONE key pair (kyber_batch_enc_shared_pk / dec_shared_sk)")
    ap.add_argument("--workload", default="kyber768", choices=sorted(WORKLOADS) + ["ntt"],
                    help="default kyber768 = BASELINE.json's metric; others are side measurements")
    args = ap.parse_args()
    if args.workload == "ntt":
        return bench_ntt(args)
    global K, N90, ENCDEC_BYTES, ENCDEC_PERMS
    K, N90, ENCDEC_BYTES, ENCDEC_PERMS = WORKLOADS[args.workload]
    set_name = "Kyber%d%s" % (256 * K, "-90s" if N90 else "")

    import numpy as np
    import torch
    import kyber_amd
    from kyber_amd import lib as kl

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torchrun, also with one rank, so the RCCL path can be rehearsed on one GPU
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    dev = torch.device("cuda", local)
    kem = kyber_amd.KyberHip(K, N90)
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
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
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
    if rank == 0:
        kl.profile_enable(True)
        reps = 3
        for _ in range(reps):
            step()
        prof = kl.profile_read()
        kl.profile_enable(False)
        tot = sum(ms for _, ms in prof.values())
        for name, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
            kernels[name] = {"launches": cnt, "avg_ms": ms / cnt, "share": ms / tot}
        dom = max(prof.items(), key=lambda kv: kv[1][1])[0]
        cnt, ms = prof[dom]
        avg_s = ms / cnt * 1e-3
        chunk = min(n, args.chunk or DEFAULT_CHUNK)
        if dom in ("k_sample_matrix", "k90_sample_matrix"):
            entries = chunk * K * K
            ops = entries * MATRIX_ENTRY_PERMS * KECCAK_LANE_OPS
            bytes_ = entries * MATRIX_ENTRY_BYTES
        else:  # fall back to whole-pipeline accounting for an unexpected dominant kernel
            ops = chunk * ENCDEC_PERMS * KECCAK_LANE_OPS
            bytes_ = chunk * ENCDEC_BYTES
        ach = ops / avg_s * 1e-12
        traffic, traffic_src = pmc_traffic(dom, K, chunk)
        roof = {
            "kernel": dom, "bound": "valu", "achieved": ach, "peak": VALU_PEAK_TLOPS, "unit": "Tlane-op/s",
            "frac": ach / VALU_PEAK_TLOPS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes": bytes_, "avg_launch_ms": avg_s * 1e3,
            "units_per_launch": chunk * K * K if dom.endswith("sample_matrix") else chunk,
            "hbm": {"achieved": bytes_ / avg_s * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": bytes_ / avg_s * 1e-9 / HBM_PEAK_GBS},
        }

    if rank == 0:
        total = n * world * args.steps
        value = total / dt
        out = {
            "metric": "%s encaps+decaps/sec%s" % (set_name, " (one shared key pair)" if args.shared_key else ""), "value": value, "unit": "encaps+decaps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (Keccak lanes as 2x32 bit) + f32 (exact integer polynomial arithmetic)",
            "data": "synthetic (device-generated coins, seed 0x4B59424552)",
            "config": {"workload": "%s batched encaps+decaps%s, %d instances per GPU per step, device-resident"
                       % (set_name, " with one shared key pair" if args.shared_key else "", n),
                       "instances_per_gpu": n, "parallelism": "independent instances sharded per GPU, no collective"},
            "parity": {"ss_dec_equals_ss_enc": True, "strided_sample_vs_cpu_reference": parity},
            "pipeline": {"valu_frac": value / world * ENCDEC_PERMS * KECCAK_LANE_OPS * 1e-12 / VALU_PEAK_TLOPS,
                         "hbm_frac": value / world * ENCDEC_BYTES * 1e-9 / HBM_PEAK_GBS},
            "roofline": roof, "kernels": kernels,
        }
        if N90:  # the Keccak lane-op constant does not describe the AES/SHA-2 sets
            out["pipeline"]["valu_frac"] = None
            if roof:
                roof.update({"achieved": None, "frac": None})
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


if __name__ == "__main__":
    main()
