#!/usr/bin/env python
"""BASELINE.json configs[4]: the device-resident ANYmal pipeline (PDAdapter -> PDController -> batched
physics -> Mahony filter, jiminy_amd.envs.make_anymal_env) driving a PyTorch PPO learner, one process
per GPU.  Observations never leave the device: the rollout buffer, the policy and the update all
live next to the physics state; the only collective is DDP's gradient all-reduce (RCCL over xGMI),
overlapped with the backward pass -- no observation gather.

    python examples/ppo_anymal.py --envs 4096 --iters 5                       # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/ppo_anymal.py --envs 8192

Prints one JSON line per run (env-steps/s with the learner in the loop).  The learner is a caller of
the hot path, not part of it: plain PyTorch modules, nothing hand-written here."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from typing import Any, Callable, Dict, Tuple

import torch
import torch.distributed as dist
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class ActorCritic(nn.Module):
    def __init__(self, obs_dim: int, act_dim: int, hidden: int = 256) -> None:
        super().__init__()
        self.pi = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                                nn.Linear(hidden, act_dim))
        self.vf = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                                nn.Linear(hidden, 1))
        self.log_std = nn.Parameter(torch.full((act_dim,), -1.0))

    def forward(self, obs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return self.pi(obs), self.log_std.expand(obs.shape[0], -1), self.vf(obs).squeeze(-1)


def gaussian_log_prob(mean: torch.Tensor, log_std: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    return (-0.5 * ((x - mean) / log_std.exp()) ** 2 - log_std - 0.9189385332046727).sum(-1)


def compute_gae(rew: torch.Tensor, val: torch.Tensor, done: torch.Tensor, last_val: torch.Tensor,
                gamma: float, lam: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """Generalised advantage estimation over a `[T][B]` rollout; `done[t]` cuts the bootstrap."""
    T = rew.shape[0]
    adv = torch.zeros_like(rew)
    nxt, acc = last_val, torch.zeros_like(last_val)
    for t in range(T - 1, -1, -1):
        nd = 1.0 - done[t]
        delta = rew[t] + gamma * nxt * nd - val[t]
        acc = delta + gamma * lam * nd * acc
        adv[t] = acc
        nxt = val[t]
    return adv, adv + val


class PPO:
    """Clipped-surrogate PPO on a vectorised, device-resident environment.

    `env_step(action) -> (obs, reward, done)` and `obs` are `[B][*]` float32 tensors on `device`;
    with an initialised process group the model is wrapped in DistributedDataParallel."""

    def __init__(self, obs_dim: int, act_dim: int, device: torch.device, lr: float = 3e-4, gamma: float = 0.99,
                 lam: float = 0.95, clip: float = 0.2, epochs: int = 2, minibatches: int = 4, vf_coef: float = 0.5,
                 ent_coef: float = 0.0, seed: int = 0) -> None:
        torch.manual_seed(seed)   # same initial weights on every rank
        self.net: nn.Module = ActorCritic(obs_dim, act_dim).to(device)
        self.model: nn.Module = self.net
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.nn.parallel import DistributedDataParallel as DDP
            self.model = DDP(self.net, device_ids=[device.index] if device.type == "cuda" else None)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr)
        self.gamma, self.lam, self.clip, self.epochs, self.minibatches = gamma, lam, clip, epochs, minibatches
        self.vf_coef, self.ent_coef, self.device = vf_coef, ent_coef, device
        self.gen = torch.Generator(device=device).manual_seed(seed + 1000 * (dist.get_rank() if dist.is_initialized() else 0))

    @torch.no_grad()
    def act(self, obs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        mean, log_std, val = self.net(obs)
        act = mean + log_std.exp() * torch.randn(mean.shape, generator=self.gen, device=mean.device)
        return act, gaussian_log_prob(mean, log_std, act), val

    def rollout(self, obs: torch.Tensor, env_step: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]],
                horizon: int) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
        B = obs.shape[0]
        buf = {k: torch.empty((horizon, B) + s, device=self.device) for k, s in
               (("obs", obs.shape[1:]), ("act", (self.net.log_std.shape[0],)), ("logp", ()), ("val", ()),
                ("rew", ()), ("done", ()))}
        for t in range(horizon):
            act, logp, val = self.act(obs)
            buf["obs"][t], buf["act"][t], buf["logp"][t], buf["val"][t] = obs, act, logp, val
            obs, rew, done = env_step(act)
            buf["rew"][t], buf["done"][t] = rew, done.float()
        with torch.no_grad():
            last_val = self.net(obs)[2]
        buf["adv"], buf["ret"] = compute_gae(buf["rew"], buf["val"], buf["done"], last_val, self.gamma, self.lam)
        return buf, obs

    def update(self, buf: Dict[str, torch.Tensor]) -> Dict[str, float]:
        flat = {k: v.reshape((-1,) + v.shape[2:]) for k, v in buf.items()}
        n = flat["obs"].shape[0]
        adv = flat["adv"]
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        stats = {"loss": 0.0, "pi_loss": 0.0, "vf_loss": 0.0}
        steps = 0
        for _ in range(self.epochs):
            perm = torch.randperm(n, generator=self.gen, device=self.device)
            for idx in perm.chunk(self.minibatches):
                mean, log_std, val = self.model(flat["obs"][idx])
                logp = gaussian_log_prob(mean, log_std, flat["act"][idx])
                ratio = (logp - flat["logp"][idx]).exp()
                a = adv[idx]
                pi_loss = -torch.min(ratio * a, ratio.clamp(1 - self.clip, 1 + self.clip) * a).mean()
                vf_loss = 0.5 * (val - flat["ret"][idx]).pow(2).mean()
                ent = (log_std + 1.4189385332046727).sum(-1).mean()
                loss = pi_loss + self.vf_coef * vf_loss - self.ent_coef * ent
                self.opt.zero_grad(set_to_none=True)
                loss.backward()        # DDP: bucketed gradient all-reduce overlapped with backward
                nn.utils.clip_grad_norm_(self.model.parameters(), 1.0)
                self.opt.step()
                stats["loss"] += float(loss.detach()); stats["pi_loss"] += float(pi_loss.detach())
                stats["vf_loss"] += float(vf_loss.detach()); steps += 1
        return {k: v / steps for k, v in stats.items()}


def flatten_anymal_obs(obs: Dict[str, Any]) -> torch.Tensor:
    """Device-side feature vector: joint positions / all velocities, IMU, Mahony quaternion, PD targets."""
    q, v = obs["states"]["agent"]["q"], obs["states"]["agent"]["v"]
    parts = [q[:, 7:], v, obs["measurements"]["ImuSensor"].flatten(1), obs["features"]["mahony_filter"].flatten(1),
             obs["actions"]["pd_controller"].flatten(1)]
    return torch.cat(parts, dim=1).float()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--minibatches", type=int, default=4)
    ap.add_argument("--contact-model", default="spring_damper", choices=["spring_damper", "constraint"],
                    help="'constraint' = the contact model of the reference's shipped ANYmal options")
    ap.add_argument("--std-ratio-ground", type=float, default=0.0,
                    help="ground-friction randomisation per environment (constraint contact model)")
    ap.add_argument("--std-ratio-sensors", type=float, default=0.0,
                    help="sensor noise / bias / delay randomisation (drawn per reset for the batch)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl")     # RCCL
    rank = dist.get_rank() if world > 1 else 0
    from jiminy_amd.envs import make_anymal_env
    std_ratio = {k: v for k, v in (("ground", args.std_ratio_ground), ("sensors", args.std_ratio_sensors)) if v > 0}
    env = make_anymal_env(args.envs, device=dev, contact_model=args.contact_model, std_ratio=std_ratio or None)
    obs_d, _ = env.reset(seed=rank)
    obs = flatten_anymal_obs(obs_d)
    ppo = PPO(obs.shape[1], env.model.nmotors, dev, epochs=args.epochs, minibatches=args.minibatches)

    def env_step(action: torch.Tensor):
        o, r, term, trunc, _ = env.step(0.25 * torch.tanh(action).double())
        return flatten_anymal_obs(o), r.float(), term | trunc
    hist = []
    t_roll = t_upd = 0.0
    for it in range(args.iters + 1):          # iteration 0 = warm-up, not timed
        torch.cuda.synchronize(); t0 = time.perf_counter()
        buf, obs = ppo.rollout(obs, env_step, args.horizon)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        st = ppo.update(buf)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it > 0:
            t_roll += t1 - t0; t_upd += t2 - t1
        hist.append((float(buf["rew"].mean()), st["loss"]))
    if world > 1:
        t = torch.tensor([t_roll, t_upd], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_roll, t_upd = float(t[0]), float(t[1])
    if rank == 0:
        steps = args.envs * world * args.horizon * args.iters
        print(json.dumps({"metric": "gym-steps/s ANYmal pipeline + PPO learner", "value": steps / (t_roll + t_upd),
                          "n_gpus": world, "envs_per_gpu": args.envs, "horizon": args.horizon, "iters": args.iters,
                          "rollout_s": t_roll, "update_s": t_upd, "rollout_only_steps_per_s": steps / t_roll,
                          "mean_reward_first_last": [hist[0][0], hist[-1][0]], "loss_first_last": [hist[0][1], hist[-1][1]],
                          "obs_dim": int(obs.shape[1]), "collective": "DDP gradient all-reduce only" if world > 1 else "none"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
