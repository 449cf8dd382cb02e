"""bench.py --gpus N starts N ranks by itself (the driver's N > 1 invocation
outside torchrun) and also runs as one rank under torchrun.  CPU rehearsal:
gloo, a stand-in engine (tests/bench_stub_engine.py), two ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["KYBER_BENCH_ENGINE"] = "bench_stub_engine:make"
    return env


def _line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_launches_two_ranks():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
           "--device", "cpu", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["dist_backend"] == "gloo"
    # whole-job value: both ranks' instances over the slowest rank's time
    assert abs(d["value"] - 2 * 16 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6


def test_failed_rank_ends_the_job_promptly():
    """one rank exits non-zero: the launcher stops the others instead of sitting out the rendezvous /
    barrier time-out, prints no JSON line, names the rank and passes its stderr on"""
    import time
    env = _env()
    env["KYBER_BENCH_ENGINE"] = "bench_stub_engine:make_failing_rank1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8",
           "--device", "cpu", "--no-cpu-baseline"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 7, (r.returncode, r.stderr.decode())
    assert time.time() - t0 < 120
    assert b"{" not in r.stdout and b"rank 1 exited with status 7" in r.stderr


def test_one_rank_under_torchrun_and_world_mismatch():
    base = [os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--batch", "8", "--device", "cpu", "--no-cpu-baseline"]
    run = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())]
    r = subprocess.run(run + base + ["--gpus", "2"], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert _line(r.stdout)["n_gpus"] == 2
    # a world that differs from --gpus is refused instead of printing a mislabelled line
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable] + base + ["--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"{" not in r.stdout


def test_cpu_device_without_stub_is_refused():
    env = _env()
    del env["KYBER_BENCH_ENGINE"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--device", "cpu", "--batch", "4"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"no CPU path" in r.stderr


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_rehearsal():
    """One-GPU box: two ranks, both on device 0 (rendezvous over gloo because
    RCCL refuses two ranks on one device); the HIP engine runs in both."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "65536",
           "--ranks-share-device", "--dist-backend", "gloo", "--no-cpu-baseline"]  # 65536: encapsulation's front stages are one launch
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["parity"]["ss_dec_equals_ss_enc"] is True
    assert d["roofline"]["kernel"] in ("k_sample_matrix", "k_enc_front", "k_dec_middle") and 0 < d["roofline"]["frac"] <= 1
