"""Synthetic state sampler and multi-process sharding (gloo, world_size 2, CPU)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jiminy_amd import load_builtin
from jiminy_amd.distributed import ObservationGather, all_gather_observations, pack_observations, shard_range
from jiminy_amd.synthetic import lowest_contact_height, sample_states


def test_sampled_states_are_valid_start_states():
    m = load_builtin("anymal")
    st = sample_states(m, 512, seed=0)
    q, v, cmd = st["q"], st["v"], st["command"]
    assert q.shape == (19, 512) and v.shape == (18, 512) and cmd.shape == (12, 512)
    assert np.allclose(np.linalg.norm(q[3:7], axis=0), 1.0, atol=1e-14)
    mask = m.bounded_position_mask()
    assert (q[mask] >= m.position_lower[mask, None]).all() and (q[mask] <= m.position_upper[mask, None]).all()
    z = lowest_contact_height(m, q)
    assert z.min() > -0.0051                      # initial contact force stays far below 1e5 N
    assert (np.abs(z[:128]) <= 0.0051).all()      # grounded quarter of the batch
    assert (np.abs(cmd) <= 40.0).all()
    st2 = sample_states(m, 512, seed=0)
    assert np.array_equal(st2["q"], q)
    assert not np.array_equal(sample_states(m, 512, seed=1)["q"], q)


def test_cartpole_sampling_range():
    m = load_builtin("cartpole")
    st = sample_states(m, 256, seed=0)
    assert np.abs(st["q"][0]).max() <= 0.05 and np.abs(st["v"]).max() <= 0.05
    assert np.allclose(st["q"][1] ** 2 + st["q"][2] ** 2, 1.0)


def test_shard_range_partitions_the_batch():
    for B, W in ((65536, 8), (32768, 8), (10, 3), (5, 8)):
        r = [shard_range(B, k, W) for k in range(W)]
        assert r[0][0] == 0 and r[-1][1] == B
        assert all(r[k][1] == r[k + 1][0] for k in range(W - 1))
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = 6
        lo, hi = shard_range(2 * B, rank, world)
        imu = torch.arange(6 * B, dtype=torch.float64).reshape(6, B) + 1000 * rank
        enc = torch.arange(4 * B, dtype=torch.float64).reshape(4, B) - 1000 * rank
        out = all_gather_observations([imu, enc])
        out = all_gather_observations([imu, enc], out)        # buffer reuse path
        packed, gathered = out
        ok = tuple(gathered.shape) == (world, 10, B) and torch.equal(packed, pack_observations([imu, enc]))
        for r in range(world):
            ref = torch.cat([torch.arange(6 * B, dtype=torch.float64).reshape(6, B) + 1000 * r,
                             torch.arange(4 * B, dtype=torch.float64).reshape(4, B) - 1000 * r])
            ok = ok and torch.equal(gathered[r], ref)
        ok = ok and (lo, hi) == (rank * B, (rank + 1) * B)
        # asynchronous, double-buffered form (what bench.py --gather-obs uses): three launches in flight order,
        # the result of the last one is the last block that was packed
        g = ObservationGather()
        for k in range(3):
            g.launch([imu + k, enc])
        last = g.result()
        for r in range(world):
            ok = ok and torch.equal(last[r][:6], torch.arange(6 * B, dtype=torch.float64).reshape(6, B) + 1000 * r + 2)
        g.drain()
        # packed in float32 (half the bytes over the links): values that are exact in float32 round-trip exactly, the
        # others to single precision; gathered every 2nd launch only: calls in between return at once and result()
        # keeps the last gathered block
        g32 = ObservationGather(dtype=torch.float32, every=2)
        started = [g32.launch([imu + k + 0.1, enc]) for k in range(4)]      # gathers k = 0 and k = 2
        last32 = g32.result()
        ok = ok and started == [True, False, True, False] and g32.launched == 2
        ok = ok and last32.dtype == torch.float32 and g32.bytes_per_rank == 10 * B * 4
        for r in range(world):
            want = torch.arange(6 * B, dtype=torch.float64).reshape(6, B) + 1000 * r + 2 + 0.1
            ok = ok and torch.equal(last32[r][:6], want.to(torch.float32))
            ok = ok and float((last32[r][:6].double() - want).abs().max()) < 1e-3
            ok = ok and torch.equal(last32[r][6:].double(), torch.arange(4 * B, dtype=torch.float64).reshape(4, B) - 1000 * r)
        g32.drain()
        # ragged shards are refused up front instead of hanging inside the collective
        try:
            all_gather_observations([imu[:, : B - rank], enc[:, : B - rank]])
            ok = False
        except ValueError:
            pass
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_observation_all_gather_world_size_2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}


def test_bench_launcher_spawns_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` without a launcher around it must start 2 ranks by itself, check the
    process-group size and print ONE JSON line from rank 0 (dry run: gloo on CPU tensors, no physics)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--gather-obs",
                          "--model", "atlas", "--batch", "32768", "--strong", "--steps", "4"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["n_ranks_rccl"] == 2 and rec["gather_ok"] and rec["scaling"] == "strong"
    # a world size that contradicts --gpus is refused, not silently run on one rank
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--dry-run"],
                         capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="2", RANK="0"))
    assert bad.returncode != 0
