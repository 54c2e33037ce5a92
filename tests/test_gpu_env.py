"""Vectorised environment surface on the GPU (reset / step / observe, device-resident)."""
import numpy as np
import pytest
import torch

from jiminy_amd.envs import PDControlledWalkerVecEnv, make_anymal_env

pytestmark = pytest.mark.gpu


def test_anymal_pd_pipeline_stands(gpu_device):
    """≙ the reference's PD standing checks (gym_jiminy/unit_py/test_pipeline_control.py:46-113):
    zero action = hold the neutral posture; the robot must stay up and come to rest."""
    B = 256
    env = make_anymal_env(B, dt_max=5e-4)
    assert isinstance(env, PDControlledWalkerVecEnv)
    obs, info = env.reset(seed=0)
    assert obs["states"]["agent"]["q"].shape == (B, 19) and obs["states"]["agent"]["q"].is_cuda
    assert set(obs["measurements"]) == {"ImuSensor", "ForceSensor", "EncoderSensor", "EffortSensor"}
    assert obs["measurements"]["ImuSensor"].shape == (B, 6, 1)
    assert obs["measurements"]["ForceSensor"].shape == (B, 6, 4)
    assert obs["measurements"]["EncoderSensor"].shape == (B, 2, 12)
    z0 = obs["states"]["agent"]["q"][:, 2].clone()
    action = torch.zeros((B, 12), dtype=torch.float64, device=gpu_device)
    for _ in range(75):   # 3 s
        obs, reward, terminated, truncated, info = env.step(action)
        assert not bool(terminated.any()) and not bool(truncated.any())
    q, v = obs["states"]["agent"]["q"], obs["states"]["agent"]["v"]
    assert float((q[:, 2] - z0).abs().max()) < 0.05
    assert float(v.abs().max()) < 1e-2
    assert float(reward.min()) == 1.0
    # the four feet carry the weight (force sensor z in the foot frame, spring-damper ground)
    fz = obs["measurements"]["ForceSensor"][:, 2, :].sum(dim=1)
    assert torch.allclose(fz, torch.full_like(fz, 52.13485 * 9.81), rtol=2e-2)
    # Mahony estimate of the trunk attitude stays close to the true (upright) one
    quat = obs["features"]["mahony_filter"][:, :, 0]
    assert float((quat[:, :3]).abs().max()) < 5e-2
    # all lanes are identical copies: bitwise repeatability across lanes (reference
    # test_pipeline_control.py:315-330 asserts it across resets)
    a = env.engine.robot_state.a
    assert bool((a == a[:, :1]).all())


def test_auto_reset_of_failed_lanes(gpu_device):
    """Without a controller the limp robot collapses: lanes terminate (fall detection) or are
    truncated (joint leaves its bounds -> per-lane status) and are re-initialised in place."""
    B = 64
    env = make_anymal_env(B, pd_pipeline=False)
    obs0, _ = env.reset(seed=0)
    q_neutral = obs0["states"]["agent"]["q"][0].clone()
    action = torch.zeros((B, 12), dtype=torch.float64, device=gpu_device)
    action[: B // 2] = 20.0
    n_reset = 0
    for _ in range(40):
        obs, reward, terminated, truncated, info = env.step(action)
        if "reset_mask" in info:
            m = info["reset_mask"]
            assert bool(((terminated | truncated) == m).all())
            n_reset += int(m.sum())
            assert float(obs["t"][m].abs().max()) == 0.0              # fresh clock
            assert bool((obs["states"]["agent"]["q"][m] == q_neutral).all())   # neutral state again
            assert int(env.engine.status[m].abs().sum()) == 0
    assert n_reset >= B // 2


def test_pd_env_recovers_from_a_nan_lane(gpu_device):
    """A lane that fails numerically (JM_LANE_NAN) poisons its controller / observer state with NaN; the
    auto-reset must hand back a clean lane (reset by selection: NaN * 0 would stay NaN), and the lane
    must keep working afterwards."""
    B = 32
    env = make_anymal_env(B, dt_max=5e-4)
    env.reset(seed=0)
    action = torch.zeros((B, 12), dtype=torch.float64, device=gpu_device)
    env.step(action)
    env.engine.field("v")[7, 5] = float("nan")
    obs, reward, terminated, truncated, info = env.step(action)
    assert bool(truncated[5]) and int(truncated.sum()) == 1
    assert bool(info["reset_mask"][5])
    flat = torch.cat([obs["states"]["agent"]["q"], obs["states"]["agent"]["v"], obs["features"]["mahony_filter"].flatten(1),
                      obs["actions"]["pd_controller"].flatten(1)] + [m.flatten(1) for m in obs["measurements"].values()], dim=1)
    assert bool(torch.isfinite(flat).all())
    assert bool(torch.isfinite(env.engine.field("command")).all())
    for _ in range(3):
        obs, reward, terminated, truncated, info = env.step(action)
        assert not bool(truncated.any()) and not bool(terminated.any())
    assert bool(torch.isfinite(obs["features"]["mahony_filter"]).all())
    # the re-initialised lane follows the same trajectory as a lane of a fresh episode would: compare with lane 0
    # three steps after ITS start (all lanes are identical copies under a zero action)
    # (a lane re-initialised INSIDE a running simulation continues with `dtMax` steps; a fresh simulation opens with the
    # reference's microsecond step, which is a property of `Engine::start`, not of the lane: DESIGN.md section 1)
    env2 = make_anymal_env(B, dt_max=5e-4)
    env2.reset(seed=0)
    env2.engine._opening_step = False
    for _ in range(3):
        obs2, *_ = env2.step(action)
    assert torch.allclose(obs["states"]["agent"]["q"][5], obs2["states"]["agent"]["q"][0], rtol=0, atol=1e-12)
    # reference constants of ANYmalPDControlJiminyEnv (gym_jiminy/envs/anymal.py:13-24, 93-96)
    assert env.simulation_duration_max == 20.0
    assert float(env.command_state_upper[1].max()) == 4.0 and float(env.command_state_upper[2].max()) == 30.0


def test_hip_pipeline_blocks_match_the_tensor_programs(gpu_device, monkeypatch):
    """`jm_block_pd_controller` / `jm_block_mahony_filter` (one HIP launch each) against the tensor
    programs of jiminy_amd/blocks.py (themselves pinned to the scalar restatement of the reference's
    numba kernels by tests/test_blocks.py): random command states that hit the position / velocity
    / acceleration bounds, random IMU data, two IMUs, still and moving lanes."""
    from jiminy_amd import blocks, load_builtin
    from jiminy_amd.engine import BatchedEngine
    model = load_builtin("anymal")
    B, M = 1000, model.nmotors
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)  # noqa: E731
    enc = eng.field("encoder")
    enc.copy_((rnd(*enc.shape) - 0.5) * 4.0)
    lo = torch.stack([-1.0 - rnd(M), -5.0 - rnd(M), -50.0 - 50 * rnd(M)])
    hi = torch.stack([1.0 + rnd(M), 5.0 + rnd(M), 50.0 + 50 * rnd(M)])
    kp, kd, lim = 100 + 1000 * rnd(M), 0.01 + 0.1 * rnd(M), 20 + 60 * rnd(M)
    cs = torch.stack([(rnd(M, B) - 0.5) * 2.6, (rnd(M, B) - 0.5) * 13, (rnd(M, B) - 0.5) * 250]).to(gpu_device)
    enc_idx = torch.randperm(M, generator=g)
    hb = blocks.HipBlocks(eng, enc_idx, lo, hi, kp, kd, lim)
    dev = lambda x: x.to(gpu_device)  # noqa: E731
    # One application from identical inputs per comparison (the ZOH integrator is discontinuous --
    # trunc(), bound activations --, so round-off would be amplified by chaining two independent
    # trajectories); the inputs of application i+1 are the tensor program's outputs of application i.
    for dt in (5e-3, 5e-3, 5e-3, 0.0):
        cs_ref, cs_hip = cs.clone(), cs.clone()
        out_ref = torch.zeros(M, B, dtype=torch.float64, device=gpu_device)
        out_hip = torch.zeros_like(out_ref)
        encv = enc.view(M, 2, B).permute(1, 0, 2)[:, enc_idx.to(gpu_device)]
        blocks.pd_controller(encv, cs_ref, dev(lo), dev(hi), dev(kp), dev(kd), dev(lim), dt, out_ref)
        hb.pd_controller(cs_hip, dt, out_hip)
        torch.cuda.synchronize()
        assert float((cs_ref - cs_hip).abs().max()) <= 1e-14 * float(cs_ref.abs().max())
        assert float((out_ref - out_hip).abs().max()) <= 1e-14 * float(out_ref.abs().max())
        assert float((cs_ref - cs).abs().max()) > 0 or dt == 0.0
        cs = cs_ref
    # Mahony: ANYmal has one IMU; emulate the engine's raw field with two IMUs through a view
    imu = eng.field("imu")
    imu.copy_((rnd(*imu.shape) - 0.5) * torch.tensor([1, 1, 1, 20, 20, 20], dtype=torch.float64)[:, None])
    imu[:, ::7] = 0.0    # still lanes: cf == 0 -> early return of the reference
    n_imu = 1
    quat = torch.nn.functional.normalize(rnd(4, n_imu, B) - 0.5, dim=0).to(gpu_device)
    bias = ((rnd(3, n_imu, B) - 0.5) * 0.1).to(gpu_device)
    bias[:, :, ::7] = 0.0
    ref = [quat.clone(), torch.zeros_like(bias), torch.zeros_like(bias), bias.clone()]
    hip = [quat.clone(), torch.zeros_like(bias), torch.zeros_like(bias), bias.clone()]
    v = imu.view(n_imu, 6, B).permute(1, 0, 2)
    for _ in range(4):
        for a, b in zip(ref, hip):
            b.copy_(a)
        blocks.mahony_filter(ref[0], ref[1], ref[2], v[:3], v[3:], ref[3], 1.0, 0.1, 5e-3)
        hb.mahony_filter(hip[0], hip[1], hip[2], hip[3], 1.0, 0.1, 5e-3)
        torch.cuda.synchronize()
        for a, b in zip(ref, hip):
            assert float((a - b).abs().max()) <= 1e-14 * max(float(a.abs().max()), 1.0)
    assert bool((hip[0][:, :, ::7] == quat[:, :, ::7]).all())   # still lanes untouched


def test_hip_pipeline_blocks_match_the_oracle(gpu_device):
    """`jm_block_pd_controller` / `jm_block_mahony_filter` on the device DIRECTLY against
    oracle/blocks_numpy.py -- the scalar, statement-by-statement restatement of the reference's numba
    kernels (proportional_derivative_controller.py:22-163, mahony_filter.py:28-95), one environment at a
    time like the reference -- with nothing of jiminy_amd/blocks.py in between.  Random command states
    that hit the position / velocity / acceleration bounds, shuffled encoder order, still and moving
    IMUs; each application starts from identical inputs (the ZOH integrator is discontinuous)."""
    from jiminy_amd import blocks, load_builtin
    from jiminy_amd.engine import BatchedEngine
    from oracle import blocks_numpy as orc
    model = load_builtin("anymal")
    B, M = 384, model.nmotors
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    rg = np.random.default_rng(11)
    enc = (rg.random((M, 2, B)) - 0.5) * 4.0           # raw encoder field [n_enc][2][B]
    eng.field("encoder").copy_(torch.from_numpy(enc.reshape(2 * M, B)))
    lo = np.stack([-1.0 - rg.random(M), -5.0 - rg.random(M), -50.0 - 50 * rg.random(M)])
    hi = np.stack([1.0 + rg.random(M), 5.0 + rg.random(M), 50.0 + 50 * rg.random(M)])
    kp, kd, lim = 100 + 1000 * rg.random(M), 0.01 + 0.1 * rg.random(M), 20 + 60 * rg.random(M)
    cs = np.stack([(rg.random((M, B)) - 0.5) * 2.6, (rg.random((M, B)) - 0.5) * 13, (rg.random((M, B)) - 0.5) * 250])
    enc_idx = rg.permutation(M)
    hb = blocks.HipBlocks(eng, torch.from_numpy(enc_idx), torch.from_numpy(lo), torch.from_numpy(hi),
                          torch.from_numpy(kp), torch.from_numpy(kd), torch.from_numpy(lim))
    n_bound_hits = 0
    for dt in (5e-3, 5e-3, 5e-3, 0.0):
        cs_dev = torch.from_numpy(cs).to(gpu_device)
        out_dev = torch.zeros((M, B), dtype=torch.float64, device=gpu_device)
        hb.pd_controller(cs_dev, dt, out_dev)
        cs_ref, out_ref = cs.copy(), np.zeros((M, B))
        for lane in range(B):
            state = np.ascontiguousarray(cs_ref[:, :, lane])
            o = np.zeros(M)
            orc.pd_controller(enc[enc_idx, :, lane].T, state, lo, hi, kp, kd, lim, dt, o)
            cs_ref[:, :, lane], out_ref[:, lane] = state, o
        got_cs, got_out = cs_dev.cpu().numpy(), out_dev.cpu().numpy()
        assert np.abs(got_cs - cs_ref).max() <= 1e-13 * np.abs(cs_ref).max()
        assert np.abs(got_out - out_ref).max() <= 1e-13 * np.abs(out_ref).max()
        n_bound_hits += int((np.abs(out_ref) == lim[:, None]).sum())
        n_bound_hits += int(((cs_ref[1] == lo[1][:, None]) | (cs_ref[1] == hi[1][:, None])).sum())
        cs = cs_ref
    assert n_bound_hits > 100   # the saturation branches are exercised
    # ---- Mahony filter (ANYmal: one IMU)
    imu = (rg.random((6, B)) - 0.5) * np.array([1, 1, 1, 20, 20, 20.0])[:, None]
    imu[:, ::7] = 0.0           # still lanes: cf == 0 -> the early return of the reference
    eng.field("imu").copy_(torch.from_numpy(imu))
    quat = rg.random((4, 1, B)) - 0.5
    quat /= np.linalg.norm(quat, axis=0, keepdims=True)
    bias = (rg.random((3, 1, B)) - 0.5) * 0.1
    bias[:, :, ::7] = 0.0
    for _ in range(4):
        dev = [torch.from_numpy(x).to(gpu_device) for x in (quat, np.zeros_like(bias), np.zeros_like(bias), bias)]
        hb.mahony_filter(dev[0], dev[1], dev[2], dev[3], 1.0, 0.1, 5e-3)
        q_ref, b_ref = quat.copy(), bias.copy()
        om_ref, cf_ref = np.zeros_like(bias), np.zeros_like(bias)
        for lane in range(B):
            q1, b1 = np.ascontiguousarray(q_ref[:, :, lane]), np.ascontiguousarray(b_ref[:, :, lane])
            om, cf = np.zeros((3, 1)), np.zeros((3, 1))
            orc.mahony_filter(q1, om, cf, imu[:3, lane][:, None], imu[3:, lane][:, None], b1, 1.0, 0.1, 5e-3)
            q_ref[:, :, lane], b_ref[:, :, lane], om_ref[:, :, lane], cf_ref[:, :, lane] = q1, b1, om, cf
        for got, want in zip(dev, (q_ref, om_ref, cf_ref, b_ref)):
            assert np.abs(got.cpu().numpy() - want).max() <= 1e-13 * max(np.abs(want).max(), 1.0)
        quat, bias = q_ref, b_ref
    assert (quat[:, :, ::7] == dev[0].cpu().numpy()[:, :, ::7]).all()


def test_env_with_hip_blocks_equals_env_with_tensor_blocks(gpu_device, monkeypatch):
    B = 128
    g = torch.Generator(device="cpu").manual_seed(1)
    actions = [(torch.rand(B, 12, generator=g, dtype=torch.float64) - 0.5).to(gpu_device) for _ in range(4)]
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("JIMINY_AMD_TENSOR_BLOCKS", flag)
        env = make_anymal_env(B, dt_max=5e-4, auto_reset=False)
        env.reset(seed=0)
        for a in actions:
            obs, *_ = env.step(a)
        outs.append((obs["states"]["agent"]["q"].clone(), obs["features"]["mahony_filter"].clone(),
                     obs["actions"]["pd_controller"].clone()))
        env.close()
    for a, b in zip(*outs):
        assert float((a - b).abs().max()) < 1e-6


def test_ppo_learner_on_the_device_resident_pipeline(gpu_device):
    """BASELINE configs[4] on one GPU: rollouts of the ANYmal pipeline feed the PPO update without
    leaving the device (examples/ppo_anymal.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import ppo_anymal as ppo
    from jiminy_amd.envs import make_anymal_env
    env = make_anymal_env(256, device=gpu_device)
    obs_d, _ = env.reset(seed=0)
    obs = ppo.flatten_anymal_obs(obs_d)
    assert obs.is_cuda and obs.dtype == torch.float32 and obs.shape == (256, 64)
    learner = ppo.PPO(obs.shape[1], env.model.nmotors, gpu_device, epochs=1, minibatches=2)

    def env_step(action):
        o, r, term, trunc, _ = env.step(0.25 * torch.tanh(action).double())
        return ppo.flatten_anymal_obs(o), r.float(), term | trunc
    for _ in range(2):
        buf, obs = learner.rollout(obs, env_step, 4)
        stats = learner.update(buf)
        assert all(v.is_cuda for v in buf.values())
        assert np.isfinite(stats["loss"]) and bool(torch.isfinite(buf["adv"]).all())
    env.close()


def test_env_sensor_randomisation(gpu_device):
    """`std_ratio={'sensors': s}` (≙ WalkerJiminyEnv, envs/locomotion.py:264-288): noise, bias, delay and
    jitter are configured for every sensor type at reset; the physics is untouched, the measurements are not,
    and a given seed reproduces the same episode."""
    def run(std_ratio, seed=7):
        env = make_anymal_env(32, device=gpu_device, std_ratio=std_ratio)
        env.reset(seed=seed)
        action = torch.zeros((32, env.model.nmotors), dtype=torch.float64, device=gpu_device)
        for _ in range(2):
            env.step(action)
        out = {k: env.engine.field(k).clone() for k in ("q", "encoder", "imu", "effort")}
        opts = {k: (None if v["delay"] is None else v["delay"].copy()) for k, v in env.engine._sensor_noise.items()}
        env.close()
        return out, opts
    clean, _ = run(None)
    noisy, opts = run({"sensors": 1.0})
    again, _ = run({"sensors": 1.0})
    assert set(opts) >= {"EncoderSensor", "ImuSensor", "EffortSensor"}
    assert opts["EncoderSensor"] is not None and 0.0 <= opts["EncoderSensor"].max() <= 3.0e-3
    assert opts["ImuSensor"] is None                      # SENSOR_DELAY_SCALE[ImuSensor] = 0
    # (the PD controller reads the measured encoders, so the motion itself differs, as in the reference)
    for k in ("encoder", "imu", "effort"):
        assert torch.isfinite(noisy[k]).all()
        assert torch.equal(noisy[k], again[k]), k
    assert not torch.equal(clean["imu"], noisy["imu"]) and not torch.equal(clean["effort"], noisy["effort"])


def test_env_model_biases_and_disturbances(gpu_device):
    """`model_options` (body biases per environment, re-drawn for the lanes being reset) and
    `std_ratio={'disturbance': s}` (impulse pushes on the root body, envs/locomotion.py:298-326)."""
    B = 64
    env = make_anymal_env(B, device=gpu_device, dt_max=5e-4,
                          model_options={"dynamics": {"massBodiesBiasStd": 0.05, "inertiaBodiesBiasStd": 0.05}},
                          std_ratio={"disturbance": 0.2})
    env.reset(seed=3)
    ml0 = env.engine.field("model_lane").clone()
    mass = ml0.view(env.model.njoints, 13, B)[2, 0]
    assert float(mass.std()) > 0.0 and abs(float(mass.mean()) / env.model.mass[2] - 1.0) < 0.05
    # the free-flyer root is not a "mechanical joint": the reference never biases its body (model.cc:337-341, 1168)
    assert float(ml0.view(env.model.njoints, 13, B)[1, 0].std()) == 0.0
    # the pushes are scheduled lazily, one period ahead at most (never one (6, B) tensor per push of the horizon)
    assert len(env.engine.impulse_forces) == 0
    # the continuous Gaussian-process force (envs/locomotion.py:327-359) is in the applied wrench from the first launch:
    # F_PROFILE_SCALE * std_ratio * process(t), x / y only, one realisation per environment
    procs = env._f_xy_profile
    w0 = env.engine.field("applied")[:6].clone()
    assert torch.allclose(w0[0], 50.0 * 0.2 * procs[0](0.0), atol=1e-9) and torch.allclose(w0[1], 50.0 * 0.2 * procs[1](0.0), atol=1e-9)
    assert float(w0[2:].abs().max()) == 0.0 and float(w0[0].std()) > 1.0
    action = torch.zeros((B, 12), dtype=torch.float64, device=gpu_device)
    z0 = env.observation()["states"]["agent"]["q"][:, :2].clone()
    seen = {}
    for _ in range(60):                      # 2.4 s: the first push has happened
        obs, _, terminated, truncated, info = env.step(action)
        imp = env.engine.impulse_forces
        assert len(imp) <= 2
        for f in imp:
            seen[round(f["t"], 9)] = f["force"]
    assert len(seen) == 1                    # period 1 only: the push of period 2 starts at 4 s +- 0.25 s
    (t1, f1), = seen.items()
    assert abs(t1 - 2.0) <= 0.25 + 1e-9
    assert 0.0 < float(f1[:2].norm(dim=0).max()) <= 0.2 * 1000.0 and float(f1[2:].abs().max()) == 0.0
    moved = (obs["states"]["agent"]["q"][:, :2] - z0).norm(dim=1)
    assert float(moved.max()) > 1e-3         # pushed sideways; the PD controller keeps them up
    assert not bool(terminated.any())
    # a reset lane gets a new biased model, the others keep theirs
    mask = torch.zeros(B, dtype=torch.bool, device=gpu_device)
    mask[:8] = True
    env.reset_lanes(mask)
    ml1 = env.engine.field("model_lane")
    assert torch.equal(ml1[:, 8:], ml0[:, 8:]) and not torch.equal(ml1[:, :8], ml0[:, :8])
    # ... and a new force process that restarts at its own episode time
    v_before = procs[0].values.clone()
    env.reset_lanes(mask)
    assert torch.equal(procs[0].values[:, 8:], v_before[:, 8:]) and not torch.equal(procs[0].values[:, :8], v_before[:, :8])
    env.step(action)
    t_lane = env._lane_time()
    assert float(t_lane[:8].max()) < float(t_lane[8:].min())
    # the schedule goes on in engine time whatever the episode length, and the lanes that were just reset are spared by
    # the push of period 2 (their episode is younger than the first push of the reference's schedule)
    for _ in range(45):                      # -> 4.3 s of engine time
        env.step(action)
        for f in env.engine.impulse_forces:
            seen[round(f["t"], 9)] = f["force"]
    assert len(seen) == 2 and len(env.engine.impulse_forces) <= 2
    f2 = seen[max(seen)]
    assert float(f2[:, :8].abs().max()) == 0.0 and float(f2[:2, 8:].norm(dim=0).max()) > 0.0


def test_hip_pd_adapter_and_motor_safety_limit_match_the_oracle(gpu_device):
    """`jm_block_pd_adapter` / `jm_block_motor_safety_limit` against oracle/blocks_numpy.py (`pd_adapter`,
    proportional_derivative_controller.py:166-260; `apply_safety_limits`, motor_safety_limit.py:20-77): both target
    orders, instantaneous or not, with and without a velocity deadband; shuffled encoder order."""
    from jiminy_amd import blocks, load_builtin
    from jiminy_amd.engine import BatchedEngine
    from oracle import blocks_numpy as orc
    model = load_builtin("anymal")
    B, M = 200, model.nmotors
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    rg = np.random.default_rng(4)
    lo = np.stack([-1.0 - rg.random(M), -3.0 - rg.random(M), -40.0 - 20 * rg.random(M)])
    hi = np.stack([1.0 + rg.random(M), 3.0 + rg.random(M), 40.0 + 20 * rg.random(M)])
    enc_idx = rg.permutation(M)
    hb = blocks.HipBlocks(eng, torch.from_numpy(enc_idx), torch.from_numpy(lo), torch.from_numpy(hi),
                          torch.ones(M, dtype=torch.float64), torch.ones(M, dtype=torch.float64),
                          torch.from_numpy(20 + 60 * rg.random(M)))
    for order in (0, 1):
        for inst in (False, True):
            for db in (None, 0.5 * rg.random(M)):
                action = (rg.random((M, B)) - 0.5) * 10.0
                cs = np.stack([(rg.random((M, B)) - 0.5) * 2, (rg.random((M, B)) - 0.5) * 6, np.zeros((M, B))])
                cs_dev = torch.from_numpy(cs).to(gpu_device)
                out_dev = torch.full((M, B), 7.0, dtype=torch.float64, device=gpu_device)
                hb.pd_adapter(torch.from_numpy(action).to(gpu_device), order, cs_dev, inst, db, 0.04, out_dev)
                cs_ref, out_ref = cs.copy(), np.full((M, B), 7.0)
                for lane in range(B):
                    state = np.ascontiguousarray(cs_ref[:, :, lane])
                    o = np.full(M, 7.0)
                    orc.pd_adapter(action[:, lane].copy(), order, state, lo, hi, inst, db, 0.04, o)
                    cs_ref[:, :, lane], out_ref[:, lane] = state, o
                assert np.abs(cs_dev.cpu().numpy() - cs_ref).max() <= 1e-13 * max(np.abs(cs_ref).max(), 1.0), (order, inst)
                assert np.abs(out_dev.cpu().numpy() - out_ref).max() <= 1e-12 * max(np.abs(out_ref).max(), 1.0), (order, inst)
    # motor safety limit
    enc = (rg.random((M, 2, B)) - 0.5) * np.array([3.0, 12.0])[None, :, None]
    eng.field("encoder").copy_(torch.from_numpy(enc.reshape(2 * M, B)))
    command = (rg.random((M, B)) - 0.5) * 200.0
    kp, kd = 20.0 + 10 * rg.random(M), 0.5 + rg.random(M)
    soft_lo, soft_hi, vlim = -1.0 - 0.2 * rg.random(M), 1.0 + 0.2 * rg.random(M), 4.0 + rg.random(M)
    out_dev = torch.zeros((M, B), dtype=torch.float64, device=gpu_device)
    hb.motor_safety_limit(torch.from_numpy(command).to(gpu_device), kp, kd, soft_lo, soft_hi, vlim, out_dev)
    out_ref = np.zeros((M, B))
    for lane in range(B):
        o = np.zeros(M)
        orc.apply_safety_limits(command[:, lane], enc[enc_idx, 0, lane], enc[enc_idx, 1, lane], kp, kd, soft_lo, soft_hi,
                                vlim, hb._lim, o)
        out_ref[:, lane] = o
    assert np.abs(out_dev.cpu().numpy() - out_ref).max() <= 1e-13 * np.abs(out_ref).max()
    assert (out_ref != command).mean() > 0.2      # the limits are active on a good part of the batch


def test_graph_replay_of_the_environment_step_is_bit_identical(gpu_device):
    """`enable_graph`: one environment step = 26 launches replayed as one captured HIP graph.  Same launches, same
    arguments: observations, rewards and the engine's time must equal the eager environment's bit for bit, across a
    lane reset and a full reset."""
    B = 256
    envs = [make_anymal_env(B, device=gpu_device, dt_max=1e-3) for _ in range(2)]
    envs[1].enable_graph()
    g = torch.Generator(device="cpu").manual_seed(0)
    for e in envs:
        e.reset(seed=5)
    for i in range(12):
        action = (0.3 * torch.randn(B, 12, generator=g, dtype=torch.float64)).to(gpu_device)
        outs = [e.step(action) for e in envs]
        for k in ("q", "v"):
            assert torch.equal(outs[0][0]["states"]["agent"][k], outs[1][0]["states"]["agent"][k]), (i, k)
        assert torch.equal(outs[0][0]["features"]["mahony_filter"], outs[1][0]["features"]["mahony_filter"])
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][3], outs[1][3])
        assert envs[0].engine.stepper_state.t == envs[1].engine.stepper_state.t
        assert envs[0].engine.stepper_state.iter == envs[1].engine.stepper_state.iter
        if i == 5:
            mask = torch.zeros(B, dtype=torch.bool, device=gpu_device)
            mask[::7] = True
            for e in envs:
                e.reset_lanes(mask)
        if i == 8:
            for e in envs:
                e.reset(seed=6)
    assert envs[1]._graph is not None
    with pytest.raises(NotImplementedError):
        e = make_anymal_env(8, device=gpu_device, dt_max=1e-3, std_ratio={"disturbance": 0.1})
        e.reset(seed=1)
        e.enable_graph()


def test_whole_environment_step_as_one_graph_matches_the_eager_step(gpu_device):
    """`enable_graph(whole_step=True)`: physics chain + episode clock + termination / truncation + reward + the masked
    auto-reset (with per-environment model biases re-drawn on the device for the lanes that restart) captured as ONE graph,
    no host read-back per step.  Wild actions make robots fall: observations, rewards, termination flags, reset masks and
    episode counters must equal the eager environment's bit for bit over many auto-resets."""
    B = 512
    mo = {"dynamics": {"massBodiesBiasStd": 0.05}}
    envs = [make_anymal_env(B, device=gpu_device, dt_max=1e-3, model_options=mo, simulation_duration_max=0.6) for _ in range(2)]
    for e in envs:
        e.reset(seed=9)
    envs[1].enable_graph(whole_step=True)
    g = torch.Generator(device="cpu").manual_seed(1)
    n_reset = 0
    for i in range(30):
        action = (6.0 * torch.randn(B, 12, generator=g, dtype=torch.float64)).to(gpu_device)
        o0, r0, te0, tr0, i0 = envs[0].step(action)
        o1, r1, te1, tr1, i1 = envs[1].step(action)
        done0 = i0.get("reset_mask", torch.zeros_like(te0))
        assert torch.equal(done0, i1["reset_mask"]), i
        n_reset += int(done0.sum())
        for k in ("q", "v"):
            assert torch.equal(o0["states"]["agent"][k], o1["states"]["agent"][k]), (i, k)
        assert torch.equal(o0["t"], o1["t"]) and torch.equal(o0["features"]["mahony_filter"], o1["features"]["mahony_filter"])
        assert torch.equal(r0, r1) and torch.equal(te0, te1) and torch.equal(tr0, tr1)
        assert torch.equal(envs[0].num_steps, envs[1].num_steps)
        assert torch.equal(envs[0].engine.field("model_lane"), envs[1].engine.field("model_lane"))
    assert n_reset > B // 4 and envs[1]._graph is not None
    # a full reset drops the graph; the next step captures it again
    for e in envs:
        e.reset(seed=10)
    assert envs[1]._graph is None
    action = torch.zeros((B, 12), dtype=torch.float64, device=gpu_device)
    # (the first step of the new simulation carries the reference's opening microsecond step: issued eagerly, the
    # periodic plan is captured at the second one)
    a, b = envs[0].step(action), envs[1].step(action)
    assert torch.equal(a[0]["states"]["agent"]["q"], b[0]["states"]["agent"]["q"]) and envs[1]._graph is None
    a, b = envs[0].step(action), envs[1].step(action)
    assert torch.equal(a[0]["states"]["agent"]["q"], b[0]["states"]["agent"]["q"]) and envs[1]._graph is not None
    with pytest.raises(NotImplementedError):
        e = make_anymal_env(8, device=gpu_device, dt_max=1e-3, contact_model="constraint", std_ratio={"ground": 0.2})
        e.reset(seed=1)
        e.enable_graph(whole_step=True)


def test_atlas_environment_step_graphs_with_the_split_constraint_stepping(gpu_device):
    """The Atlas environment (constraint contacts: step launches in the pre | solve | post form, three kernels per
    evaluation) replayed as a captured graph -- the physics chain and the whole step -- against the eager step."""
    from jiminy_amd.envs import make_atlas_env
    B = 64
    envs = [make_atlas_env(B, device=gpu_device) for _ in range(3)]
    for e in envs:
        e.reset(seed=5)
    envs[1].enable_graph()
    envs[2].enable_graph(whole_step=True)
    g = torch.Generator(device="cpu").manual_seed(2)
    n_act = envs[0].engine.model.nmotors
    for i in range(4):
        action = (0.2 * torch.randn(B, n_act, generator=g, dtype=torch.float64)).to(gpu_device)
        outs = [e.step(action) for e in envs]
        for o in outs[1:]:
            for k in ("q", "v"):
                assert torch.equal(outs[0][0]["states"]["agent"][k], o[0]["states"]["agent"][k]), (i, k)
            assert torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][2], o[2])
    assert envs[1]._graph is not None and envs[2]._graph is not None
    assert bool(torch.isfinite(envs[0].engine.field("q")).all())


def test_atlas_pd_environment_stands_with_the_reference_constants(gpu_device):
    """`make_atlas_env` ≙ `AtlasPDControlJiminyEnv` (gym_jiminy envs/atlas.py): 30 motors under MotorSafetyLimit -> PD
    controller -> PD adapter, Mahony filter, constraint contact model, Euler 1 ms / controller 5 ms.  With a zero action
    (hold the neutral pose, arms folded) the robots keep standing for a second of simulated time, the estimated attitude
    stays upright, and the HIP-graph replay of the step gives the same result."""
    from jiminy_amd.envs import ATLAS_NEUTRAL_JOINTS, make_atlas_env
    B = 64
    envs = [make_atlas_env(B, device=gpu_device) for _ in range(2)]
    envs[1].enable_graph()
    for e in envs:
        obs, _ = e.reset(seed=2)
    m = envs[0].model
    q0 = obs["states"]["agent"]["q"].clone()        # (the observation is a live view of the engine's state)
    for name, value in ATLAS_NEUTRAL_JOINTS.items():
        assert torch.allclose(q0[:, int(m.idx_q[m.joint_names.index(name)])], torch.full((B,), value, dtype=torch.float64, device=gpu_device))
    action = torch.zeros((B, m.nmotors), dtype=torch.float64, device=gpu_device)
    for _ in range(25):
        outs = [e.step(action) for e in envs]
    (obs, reward, terminated, truncated, _), (obs_g, *_rest) = outs
    assert not bool(terminated.any()) and not bool(truncated.any())
    z = obs["states"]["agent"]["q"][:, 2]
    assert float((z - q0[:, 2]).abs().max()) < 0.01                     # still standing at the neutral height
    quat = obs["features"]["mahony_filter"][:, :, 0]                      # (B, 4) first IMU
    assert float(quat[:, 3].abs().min()) > 0.95                          # estimated attitude upright
    assert torch.equal(obs["states"]["agent"]["q"], obs_g["states"]["agent"]["q"])
    assert (envs[0].engine.field("con_flags")[envs[0].engine.field("con_flags").shape[0] - m.ncontacts:] & 1).sum() >= 4 * B


@pytest.mark.parametrize("contact_model", ["spring_damper", "constraint"])
def test_env_with_every_randomisation_switched_on(gpu_device, contact_model):
    """Everything the reference's locomotion environment randomises, at once: ground friction, a random tile terrain (every
    environment on its own patch of it), sensor
    noise / bias / delay, body-parameter biases, impulse pushes and the Gaussian-process force, with random actions and
    auto-resets in flight: 40 environment steps (1.6 s) run through, observations and rewards stay finite, finished
    lanes restart, and a second run from the same seed reproduces the first bit for bit."""
    from jiminy_amd.terrain import random_tile_ground
    B = 128
    # ground friction per environment and a random tile terrain (`tiles`, random.cc:552) under BOTH contact models
    std = {"sensors": 0.3, "disturbance": 0.3, "ground": 0.5}
    terrain = (random_tile_ground((0.4, 0.4), 0.02, (0.05, 0.05), 2, 0.3, 17), (-3.0, 3.0), (-3.0, 3.0), 0.02)

    def run():
        env = make_anymal_env(B, device=gpu_device, contact_model=contact_model, std_ratio=std, ground_profile=terrain,
                              ground_patch_extent=(2.0, 2.0),
                              model_options={"dynamics": {"massBodiesBiasStd": 0.05, "centerOfMassPositionBodiesBiasStd": 0.02}})
        assert env.engine._ground is None
        env.reset(seed=11)
        assert env.engine._ground is not None and float(env.engine._ground.max()) > 0.0 and "friction" in env.engine._fields
        # every environment on its own patch of the terrain
        off = env.engine.field("ground_offset")
        assert off.shape == (2, B) and float(off.abs().max()) <= 2.0 and float(off.std()) > 0.5
        g = torch.Generator(device="cpu").manual_seed(4)
        n_reset, last = 0, None
        for i in range(40):
            action = (1.5 * torch.randn(B, 12, generator=g, dtype=torch.float64)).to(gpu_device)
            obs, reward, terminated, truncated, info = env.step(action)
            n_reset += int(info["reset_mask"].sum()) if "reset_mask" in info else 0
            for name, leaf in (("q", obs["states"]["agent"]["q"]), ("v", obs["states"]["agent"]["v"]),
                               ("mahony_filter", obs["features"]["mahony_filter"]), ("reward", reward)):
                assert bool(torch.isfinite(leaf).all()), (i, name, int((~torch.isfinite(leaf)).sum()))
            last = (obs["states"]["agent"]["q"].clone(), reward.clone())
        return n_reset, last
    n1, a = run()
    n2, b = run()
    assert n1 == n2 and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.gpu
def test_direction_reward_is_the_mean_lateral_position_of_the_episode(gpu_device):
    """`WalkerJiminyEnv.compute_reward`, 'direction' (envs/locomotion.py:419-424): at termination, minus the absolute mean
    of the free-flyer Y positions of the episode (initial state included); nothing before."""
    B = 32
    env = make_anymal_env(B, pd_pipeline=False, auto_reset=False, reward_mixture={"direction": 2.0})
    obs, _ = env.reset(seed=1)
    ys = [obs["states"]["agent"]["q"][:, 1].double().cpu().numpy().copy()]
    action = torch.zeros((B, 12), dtype=torch.float64, device=gpu_device)
    action[:, 0::3] = torch.linspace(-30.0, 30.0, B, device=gpu_device, dtype=torch.float64)[:, None]   # hip abduction: tips sideways
    done_at = np.full(B, -1)
    for step in range(40):
        obs, reward, terminated, truncated, _ = env.step(action)
        ys.append(obs["states"]["agent"]["q"][:, 1].double().cpu().numpy().copy())
        r, t = reward.cpu().numpy(), terminated.cpu().numpy()
        live = done_at < 0
        assert np.all(r[live & ~t] == 0.0)                       # no contribution before the episode ends
        for lane in np.nonzero(live & t)[0]:
            want = -2.0 * abs(np.mean([y[lane] for y in ys]))
            assert abs(r[lane] - want) <= 1e-12 * max(1.0, abs(want)), (lane, r[lane], want)
            done_at[lane] = step
    assert (done_at >= 0).sum() >= B // 4                        # the limp robots did fall


@pytest.mark.gpu
def test_flexibility_parameters_are_drawn_per_environment_and_episode(gpu_device):
    """`WalkerJiminyEnv._setup`, `std_ratio['model']` (envs/locomotion.py:288-296): every entry of `flexibilityConfig` gets
    `stiffness += 1000 sample(scale)`, `damping += 10 sample(scale)` (one draw each, the three axes alike) at every reset of
    an environment; the engine steps every environment with its own values."""
    from jiminy_amd.envs import WalkerVecEnv
    from tests import robots
    model = robots.tree_arm_flexible(True)
    B, scale = 64, 0.02      # (+- 20 N m/rad, +- 0.2 N m s/rad: below the smallest nominal values, nothing clamped at zero)
    env = WalkerVecEnv(model, B, 2e-3, engine_options={"stepper": {"odeSolver": "runge_kutta_4", "dtMax": 5e-4},
                                                      "contacts": {"model": "spring_damper"}},
                       device=gpu_device, auto_reset=False, std_ratio={"model": scale})
    env.reset(seed=3)
    flex = model.flexibility_joint_indices
    rows = env.engine.field("flexibility").cpu().numpy().reshape(len(flex), 2, 3, B)
    for i, j in enumerate(flex):
        dk = rows[i, 0] - model.flex_stiffness[j][:, None]
        dd = rows[i, 1] - model.flex_damping[j][:, None]
        assert np.abs(dk - dk[0]).max() < 1e-12 and np.abs(dd - dd[0]).max() < 1e-12     # one draw for the three axes
        assert np.abs(dk).max() <= 1000.0 * scale and np.abs(dd).max() <= 10.0 * scale
        assert dk[0].std() > 0.3 * 1000.0 * scale and dd[0].std() > 0.3 * 10.0 * scale    # ... per environment (uniform: 0.58)
    assert abs(np.corrcoef(rows[0, 0, 0], rows[1, 0, 0])[0, 1]) < 0.5                     # ... and per flexibility joint
    # the environments evolve differently from a batch that keeps the model's values
    plain = WalkerVecEnv(model, B, 2e-3, engine_options={"stepper": {"odeSolver": "runge_kutta_4", "dtMax": 5e-4},
                                                        "contacts": {"model": "spring_damper"}},
                         device=gpu_device, auto_reset=False)
    plain.reset(seed=3)
    assert torch.equal(env.engine.field("q"), plain.engine.field("q"))
    action = torch.zeros((B, model.nmotors), dtype=torch.float64, device=gpu_device)
    for _ in range(5):
        env.step(action)
        plain.step(action)
    assert (env.engine.field("q") - plain.engine.field("q")).abs().max() > 1e-6
    # a lane reset draws again for those lanes only
    mask = torch.zeros(B, dtype=torch.bool, device=gpu_device)
    mask[::2] = True
    env.reset_lanes(mask)
    rows2 = env.engine.field("flexibility").cpu().numpy().reshape(len(flex), 2, 3, B)
    assert np.array_equal(rows2[..., 1::2], rows[..., 1::2]) and not np.array_equal(rows2[..., ::2], rows[..., ::2])


@pytest.mark.gpu
def test_disturbance_forces_on_a_robot_of_the_one_robot_per_lane_family(gpu_device):
    """`std_ratio['disturbance']` (envs/locomotion.py:298-359: the Gaussian-process force profile and the impulses on the root
    body) on a robot that the one-robot-per-lane kernels step (a free-flyer with flexibility joints): the applied wrench of every
    environment reaches `f_external` of the root joint -- rotated into the joint frame (convertForceGlobalFrameToJoint) --, and
    the robots drift apart from an undisturbed batch."""
    from jiminy_amd.envs import WalkerVecEnv
    from tests import robots
    model = robots.tree_arm_flexible(True)
    B = 32
    opts = {"stepper": {"odeSolver": "runge_kutta_4", "dtMax": 5e-4}, "contacts": {"model": "spring_damper"}}
    env = WalkerVecEnv(model, B, 2e-3, engine_options=opts, device=gpu_device, auto_reset=False, std_ratio={"disturbance": 0.5})
    plain = WalkerVecEnv(model, B, 2e-3, engine_options=opts, device=gpu_device, auto_reset=False)
    env.engine.enable_output("f_external")
    plain.engine.enable_output("f_external")
    env.reset(seed=5)
    plain.reset(seed=5)
    assert torch.equal(env.engine.field("q"), plain.engine.field("q"))
    w = env.engine.field("applied")[:6].clone()                  # world-aligned (force, moment) on the root body
    assert float(w[:2].abs().max()) > 1.0 and float(w[2:].abs().max()) == 0.0
    q = env.engine.field("q")
    x, y, z, s = q[3], q[4], q[5], q[6]                          # free-flyer quaternion (x, y, z, w): world -> joint = R^T
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - s * z), 2 * (x * z + s * y)]),
                     torch.stack([2 * (x * y + s * z), 1 - 2 * (x * x + z * z), 2 * (y * z - s * x)]),
                     torch.stack([2 * (x * z - s * y), 2 * (y * z + s * x), 1 - 2 * (x * x + y * y)])])
    want = torch.einsum("ijb,ib->jb", R, w[:3])
    got = (env.engine.field("f_external") - plain.engine.field("f_external"))[6:9]      # root joint, linear part
    assert float((got - want).abs().max()) < 1e-9 * float(want.abs().max())
    action = torch.zeros((B, model.nmotors), dtype=torch.float64, device=gpu_device)
    for _ in range(10):
        env.step(action)
        plain.step(action)
    assert float((env.engine.field("q")[:2] - plain.engine.field("q")[:2]).abs().max()) > 1e-6


@pytest.mark.gpu
def test_model_biases_per_environment_on_a_robot_of_the_one_robot_per_lane_family(gpu_device):
    """`model_options` (`massBodiesBiasStd` ...: `Model::addBiasedToExtendedModel` per environment and episode) through the env
    on a robot the one-robot-per-lane kernels step: the biased models are drawn on the device, bound, and used -- heavier robots
    fall alike but load their joints differently."""
    from jiminy_amd.envs import WalkerVecEnv
    from tests import robots
    model = robots.tree_arm_flexible(True)
    B = 32
    opts = {"stepper": {"odeSolver": "runge_kutta_4", "dtMax": 5e-4}, "contacts": {"model": "spring_damper"}}
    env = WalkerVecEnv(model, B, 2e-3, engine_options=opts, device=gpu_device, auto_reset=False,
                       model_options={"dynamics": {"massBodiesBiasStd": 0.1, "inertiaBodiesBiasStd": 0.1}})
    plain = WalkerVecEnv(model, B, 2e-3, engine_options=opts, device=gpu_device, auto_reset=False)
    env.reset(seed=9)
    plain.reset(seed=9)
    ml = env.engine.field("model_lane").view(model.njoints, 13, B)
    mass = ml[2, 0]
    assert float(mass.std()) > 0.0 and abs(float(mass.mean()) / model.mass[2] - 1.0) < 0.1
    assert float(ml[1, 0].std()) == 0.0                          # the free-flyer's body is never biased (model.cc:337-341)
    action = torch.zeros((B, model.nmotors), dtype=torch.float64, device=gpu_device)
    for _ in range(5):
        env.step(action)
        plain.step(action)
    assert float((env.engine.field("v") - plain.engine.field("v")).abs().max()) > 1e-6
    assert int((env.engine.status & 1).sum()) == int((plain.engine.status & 1).sum())
