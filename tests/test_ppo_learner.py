"""The PPO learner of examples/ppo_anymal.py (BASELINE configs[4] caller) on a toy device-agnostic
environment: GAE against a direct evaluation, policy improvement, and gradient synchronisation over
a 2-process gloo group (the N > 1 path the GPU run uses with RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import ppo_anymal as ppo  # noqa: E402


def test_gae_matches_direct_sum():
    T, B, g, lam = 6, 3, 0.9, 0.8
    gen = torch.Generator().manual_seed(0)
    rew, val = torch.randn(T, B, generator=gen), torch.randn(T, B, generator=gen)
    done = (torch.rand(T, B, generator=gen) < 0.2).float()
    last = torch.randn(B, generator=gen)
    adv, ret = ppo.compute_gae(rew, val, done, last, g, lam)
    vals = torch.cat([val, last[None]])
    for b in range(B):
        for t in range(T):
            a, w = 0.0, 1.0
            for k in range(t, T):
                nd = 1.0 - float(done[k, b])
                a += w * float(rew[k, b] + g * vals[k + 1, b] * nd - vals[k, b])
                w *= g * lam * nd
                if nd == 0.0:
                    break
            assert abs(a - float(adv[t, b])) < 1e-5
    assert torch.allclose(ret, adv + val)


class _PointEnv:
    """Reward = -|x + a|^2 on a random 2-D state: the optimal policy is a = -x."""
    def __init__(self, B, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.B = B
        self.x = torch.randn(B, 2, generator=self.gen)

    def step(self, a):
        r = -((self.x + a) ** 2).sum(-1)
        self.x = torch.randn(self.B, 2, generator=self.gen)
        return self.x.clone(), r, torch.zeros(self.B, dtype=torch.bool)


def test_ppo_improves_on_a_toy_problem():
    # (one thread: in the full suite the worker threads of the host-emulation libraries loaded before this test spin next to
    # torch's, and these tiny tensor operations took two minutes instead of seven seconds)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        env = _PointEnv(256, 0)
        learner = ppo.PPO(2, 2, torch.device("cpu"), lr=3e-3, epochs=4, minibatches=2, gamma=0.0, lam=0.0)
        obs = env.x.clone()
        rewards = []
        for it in range(30):
            buf, obs = learner.rollout(obs, env.step, 8)
            learner.update(buf)
            rewards.append(float(buf["rew"].mean()))
        assert np.mean(rewards[-5:]) > np.mean(rewards[:5]) + 0.5
    finally:
        torch.set_num_threads(threads)


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = _PointEnv(64, 10 + rank)          # different data on every rank
    learner = ppo.PPO(2, 2, torch.device("cpu"), epochs=1, minibatches=2, gamma=0.0, lam=0.0)
    obs = env.x.clone()
    for _ in range(3):
        buf, obs = learner.rollout(obs, env.step, 4)
        learner.update(buf)
    flat = torch.cat([p.detach().flatten() for p in learner.net.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    q.put((rank, bool(all(torch.equal(gathered[0], g) for g in gathered)), float(flat.abs().sum())))
    dist.destroy_process_group()


def test_ddp_keeps_the_replicas_identical_gloo_world2():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
