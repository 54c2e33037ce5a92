"""Batched pipeline blocks (jiminy_amd/blocks.py, tensor programs) against the scalar
restatement of the reference's numba kernels (oracle/blocks_numpy.py)."""
import numpy as np
import pytest
import torch

from jiminy_amd import blocks
from oracle import blocks_numpy as ref


def _bounds(M, rng):
    lo = np.stack([-rng.uniform(0.3, 2.0, M), -rng.uniform(2.0, 8.0, M), -rng.uniform(50, 200, M)])
    hi = np.stack([rng.uniform(0.3, 2.0, M), rng.uniform(2.0, 8.0, M), rng.uniform(50, 200, M)])
    return lo, hi


def test_integrate_zoh_matches_reference_semantics():
    rng = np.random.default_rng(0)
    M, B, dt = 12, 64, 5e-3
    lo, hi = _bounds(M, rng)
    state = np.stack([rng.uniform(-1.5, 1.5, (M, B)), rng.uniform(-9, 9, (M, B)), rng.uniform(-300, 300, (M, B))])
    state[0] = np.clip(state[0], lo[0][:, None], hi[0][:, None])
    t = torch.from_numpy(state.copy())
    for _ in range(20):
        blocks.integrate_zoh(t, torch.from_numpy(lo), torch.from_numpy(hi), dt)
        for b in range(B):
            s = np.ascontiguousarray(state[:, :, b])
            ref.integrate_zoh(s, lo, hi, dt)
            state[:, :, b] = s
        assert np.abs(t.numpy() - state).max() < 1e-12
        state[2] = rng.uniform(-300, 300, (M, B))
        t[2] = torch.from_numpy(state[2])
    assert (state[0] <= hi[0][:, None] + 1e-9).all() and (state[0] >= lo[0][:, None] - 1e-9).all()


def test_pd_controller_and_adapter():
    rng = np.random.default_rng(1)
    M, B, dt = 12, 32, 5e-3
    lo, hi = _bounds(M, rng)
    kp, kd = rng.uniform(500, 2000, M), rng.uniform(0.005, 0.02, M)
    lim = rng.uniform(40, 80, M)
    cmd = np.stack([rng.uniform(-0.2, 0.2, (M, B)), np.zeros((M, B)), np.zeros((M, B))])
    tcmd = torch.from_numpy(cmd.copy())
    out_t = torch.zeros((M, B), dtype=torch.float64)
    acc_t = torch.zeros((M, B), dtype=torch.float64)
    db = np.full(M, 1e-2)
    for it in range(15):
        action = rng.uniform(-1.0, 1.0, (M, B))
        enc = np.stack([rng.uniform(-0.5, 0.5, (M, B)), rng.uniform(-2, 2, (M, B))])
        order, inst = it % 2, (it % 3 == 0)
        blocks.pd_adapter(torch.from_numpy(action), order, tcmd, torch.from_numpy(lo), torch.from_numpy(hi),
                          inst, torch.from_numpy(db), 0.04, acc_t)
        if not inst:
            tcmd[2].copy_(acc_t)
        blocks.pd_controller(torch.from_numpy(enc), tcmd, torch.from_numpy(lo), torch.from_numpy(hi),
                             torch.from_numpy(kp), torch.from_numpy(kd), torch.from_numpy(lim), dt, out_t)
        for b in range(B):
            cs = np.ascontiguousarray(cmd[:, :, b])
            acc = np.zeros(M)
            ref.pd_adapter(action[:, b].copy(), order, cs, lo, hi, inst, db, 0.04, acc)
            if not inst:
                cs[2] = acc
            out = np.zeros(M)
            ref.pd_controller(enc[:, :, b], cs, lo, hi, kp, kd, lim, dt, out)
            cmd[:, :, b] = cs
            assert np.abs(out_t.numpy()[:, b] - out).max() < 1e-9
        assert np.abs(tcmd.numpy() - cmd).max() < 1e-11


def test_mahony_filter_tracks_and_matches_reference():
    rng = np.random.default_rng(2)
    B, dt, kp, ki = 48, 5e-3, 1.0, 0.1
    q = np.tile(np.array([0.0, 0.0, 0.0, 1.0])[:, None], (1, B))
    bias = np.zeros((3, B))
    tq, tb = torch.from_numpy(q.copy()), torch.from_numpy(bias.copy())
    tom, tcf = torch.zeros((3, B), dtype=torch.float64), torch.zeros((3, B), dtype=torch.float64)
    for it in range(200):
        gyro = rng.normal(0, 0.3, (3, B))
        acc = np.array([0.0, 0.0, 9.81])[:, None] + rng.normal(0, 0.5, (3, B))
        if it % 50 == 0:
            gyro[:, :4] = 0.0           # lanes at rest exercise the early-return branch
            acc[:, :4] = 0.0
            bias[:, :4] = 0.0
            tb[:, :4] = 0.0
        blocks.mahony_filter(tq, tom, tcf, torch.from_numpy(gyro), torch.from_numpy(acc), tb, kp, ki, dt)
        for b in range(B):
            qq, bb = q[:, b].copy(), bias[:, b].copy()
            om, cf = np.zeros(3), np.zeros(3)
            ref.mahony_filter(qq, om, cf, gyro[:, b], acc[:, b], bb, kp, ki, dt)
            q[:, b], bias[:, b] = qq, bb
        assert np.abs(tq.numpy() - q).max() < 1e-12
        assert np.abs(tb.numpy() - bias).max() < 1e-12
    assert np.allclose(np.linalg.norm(q, axis=0), 1.0, atol=1e-9)
