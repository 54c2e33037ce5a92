"""The N > 1 path on real GPUs over RCCL: skipped on a single-GPU box, live wherever the suite runs
with several MI355X visible (the world-size-2 gloo tests of test_synthetic_and_sharding.py cover the same code on CPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus() -> int:
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


@pytest.mark.gpu
def test_observation_gather_over_rccl_equals_the_single_gpu_batch(gpu_device):
    n = _gpus()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    for world in sorted({2, n}):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                              "--master-addr", "127.0.0.1", "--master-port", str(29571 + world),
                              os.path.join(ROOT, "tests", "dist_gather_check.py")],
                             capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert line == {"world": world, "backend": "nccl", "gathered_equals_single_gpu": True}


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--gather-obs"], ["--model", "atlas", "--batch", "32768", "--strong", "--gather-obs", "--dt", "2.5e-4"]])
def test_bench_launches_one_rank_per_gpu_over_rccl(gpu_device, extra):
    """`python bench.py --gpus N` self-spawns its ranks (torch.distributed.run, rendezvous on 127.0.0.1): every rank must
    reach the barrier (`n_ranks_rccl == N`), weak scaling for ANYmal, BASELINE config 4 (`--strong`) for Atlas."""
    n = _gpus()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline", *extra], capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == n and line["n_ranks_rccl"] == n
    assert line["scaling"] == ("strong" if "--strong" in extra else "weak")
    assert line["value"] > 0.0


@pytest.mark.gpu
def test_ppo_example_runs_data_parallel_over_rccl(gpu_device):
    n = _gpus()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
                          "--master-addr", "127.0.0.1", "--master-port", "29591",
                          os.path.join(ROOT, "examples", "ppo_anymal.py"), "--envs", "1024", "--iters", "2", "--horizon", "4"],
                         capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
