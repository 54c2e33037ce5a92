"""The N > 1 path on real GPUs over RCCL: the multi-GPU cases are skipped on a single-GPU box and live
wherever the suite runs with several MI355X visible (the world-size-2 gloo tests of test_synthetic_and_sharding.py cover
the same code on CPU); the world-of-one cases at the end execute the SAME code -- `init_process_group("nccl")`, barrier,
all-reduce, the asynchronous `all_gather_into_tensor` on RCCL's stream -- on the one GPU every box has."""
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


@pytest.mark.gpu
def test_rccl_path_executes_with_a_world_of_one(gpu_device):
    """RCCL itself on this box: one rank under torch.distributed.run runs tests/dist_gather_check.py -- communicator
    creation, the shard-size all-reduce, five asynchronous all-gathers overlapped with the stepping, the final
    all-reduce -- and the gathered block must equal the single-GPU batch bit for bit."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                          "--master-addr", "127.0.0.1", "--master-port", "29561",
                          os.path.join(ROOT, "tests", "dist_gather_check.py")],
                         capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"world": 1, "backend": "nccl", "gathered_equals_single_gpu": True}


@pytest.mark.gpu
def test_bench_rank_under_torchrun_with_the_observation_gather(gpu_device):
    """The driver's N > 1 command line with N = 1 and `--gather-obs`: the rank initialises RCCL, times its steps between
    RCCL barriers, takes the max over ranks with an all-reduce and overlaps a float32 all-gather with the stepping."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                          "--master-addr", "127.0.0.1", "--master-port", "29563", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "6", "--warmup", "2", "--batch", "8192", "--no-cpu-baseline",
                          "--no-secondary", "--gather-obs", "--gather-dtype", "f32", "--gather-every", "2"],
                         capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["n_ranks_rccl"] == 1 and line["value"] > 0.0
    assert line["gather"]["collectives"] >= 3 and line["gather"]["dtype"] == "f32" and line["gather"]["bytes_per_rank_per_collective"] > 0
