"""Vectorised environments on top of BatchedEngine: the gym_jiminy reset / step / observe surface
for B lanes at once, every array a device tensor.

Restates, for the hot-path subset, `BaseJiminyEnv` (reference
python/gym_jiminy/common/gym_jiminy/common/envs/generic.py: reset :521, step :761,
`_sample_state` :1300, `has_terminated` :1434, observation layout :1247-1270) and
`WalkerJiminyEnv` (envs/locomotion.py: fall detection :361-385, survival / energy reward :387-431),
plus the ANYmal pipeline of `gym_jiminy/envs/anymal.py:82-127` (PD controller + Mahony filter) as
device-resident blocks.  Differences that come with batching are explicit: numerical failures are
per lane (`truncated`), and finished lanes are re-initialised in place (`jm_batch_reset_lanes`)
instead of raising for the whole batch.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from . import _abi, blocks
from .engine import BatchedEngine
from .model import CompiledModel, JT_FREEFLYER, load_builtin
from .synthetic import lowest_contact_height

ObsType = Dict[str, Any]


class VecJiminyEnv:
    """B independent single-robot environments stepped by one kernel launch.

    `action` is the motor command `[B][nmotors]` unless a controller block is configured.
    Observations follow the reference layout with a leading batch axis:
    `{'t': (B,), 'states': {'agent': {'q': (B, nq), 'v': (B, nv)}},
      'measurements': {SensorType: (B, n_fields, n_sensors)}}` -- zero-copy views of the
    engine's `[rows][B]` storage.
    """

    def __init__(self, model: CompiledModel, num_envs: int, step_dt: float,
                 engine_options: Optional[Dict[str, Dict[str, Any]]] = None,
                 dtype: torch.dtype = torch.float64, device: Optional[torch.device] = None,
                 simulation_duration_max: float = 86400.0, auto_reset: bool = True,
                 std_ratio: Optional[Dict[str, float]] = None,
                 model_options: Optional[Dict[str, Dict[str, float]]] = None,
                 ground_profile: Optional[Tuple[Any, Tuple[float, float], Tuple[float, float], float]] = None,
                 ground_patch_extent: Optional[Tuple[float, float]] = None) -> None:
        self.model = model
        # every environment its own patch of the ground profile: at every (lane) reset a new (x, y) offset of its
        # height-map queries, uniform in +- extent (`BatchedEngine.set_ground_offsets`) -- the batched form of a new random
        # `groundProfile` per environment instance and episode
        if ground_patch_extent is not None and ground_profile is None:
            raise ValueError("ground_patch_extent needs a ground_profile to take the patches from")
        self._ground_patch_extent = ground_patch_extent
        # ≙ `engine_options["world"]["groundProfile"]`: `(heightmap(x, y), x_range, y_range, resolution)`, e.g. a
        # `jiminy_amd.terrain.random_tile_ground` generator; sampled on the device at every `reset()`, shared by the batch
        self._ground_profile = ground_profile
        # ≙ `WalkerJiminyEnv(std_ratio=...)` (envs/locomotion.py:100-135): scale of the episode-wise
        # randomisation.  Supported keys: 'ground' (friction coefficient of every environment, constraint
        # contact model) and 'sensors' (white noise, bias, delay and jitter of every sensor type, drawn once per
        # `reset()` for the whole batch: the sensor options are lane-uniform kernel parameters) and 'disturbance'
        # (impulse forces on the root body, envs/locomotion.py:298-331: one random horizontal push per environment
        # every F_IMPULSE_PERIOD seconds of ENGINE time).  `model_options` ≙ `robot.set_model_options`: standard
        # deviations of the body mass / centre of mass / inertia / relative position biases
        # (`{"dynamics": {"massBodiesBiasStd": ...}}`, model.h:147-158); every environment gets its own biased model,
        # drawn again whenever it is reset
        self.std_ratio = dict(std_ratio or {})
        self._model_options = model_options
        self.num_envs = int(num_envs)
        self.step_dt = float(step_dt)
        self.engine = BatchedEngine(model, num_envs, dtype=dtype, device=device)
        self.device, self.dtype = self.engine.device, dtype
        if engine_options:
            self.engine.set_options(engine_options)
        if model_options:
            self.engine.set_model_options(model_options)
        self.simulation_duration_max = float(simulation_duration_max)
        self.auto_reset = auto_reset
        self._generator = torch.Generator(device="cpu")
        self.num_steps = torch.zeros(self.num_envs, dtype=torch.int64, device=self.device)
        self._t0 = torch.zeros(self.num_envs, dtype=self.dtype, device=self.device)
        # engine time on the device (set from the host clock around every step, never read back): episode times,
        # truncation and the reset bookkeeping are tensor programs that a captured graph can replay
        self._clock = torch.zeros((), dtype=self.dtype, device=self.device)
        self._state_cache: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self._q0 = None
        self._v0 = None
        q_neutral, _ = self._sample_state_numpy()
        self._height_neutral = float(q_neutral[2]) if model.has_freeflyer else 0.0

    # ------------------------------------------------------------------ overridable hooks
    def _sample_state_numpy(self) -> Tuple[np.ndarray, np.ndarray]:
        """≙ `BaseJiminyEnv._sample_state` (generic.py:1300-1334): neutral configuration clipped
        to the bounds, free-flyer placed so that the lowest contact point touches the ground."""
        m = self.model
        q = m.neutral()
        mask = m.bounded_position_mask()
        q[mask] = np.clip(q[mask], m.position_lower[mask], m.position_upper[mask])
        if m.has_freeflyer and m.ncontacts:
            q[2] -= float(lowest_contact_height(m, q)[0])
        return q, np.zeros(m.nv)

    def _sample_state(self, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Initial state for `n` lanes, `[nq][n]` and `[nv][n]`. Default: the neutral state."""
        q, v = self._sample_state_numpy()
        qt = torch.as_tensor(q, dtype=self.dtype, device=self.device)[:, None].expand(-1, n)
        vt = torch.as_tensor(v, dtype=self.dtype, device=self.device)[:, None].expand(-1, n)
        return qt.contiguous(), vt.contiguous()

    def compute_command(self, action: torch.Tensor) -> torch.Tensor:
        """≙ `BaseJiminyEnv.compute_command` (generic.py:1140-1160): the action IS the command.
        `action` is `[B][nmotors]`; returns `[nmotors][B]`."""
        return action.to(self.dtype).T

    def has_terminated(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(terminated, truncated) per lane: truncation on numerical failure / out-of-bounds
        state (generic.py:1434-1470) or on the maximum simulated duration."""
        # a PGS solve that hit its iteration cap is not fatal (the reference only counts it,
        # engine.cc:3755-3768)
        status = self.engine.status & ~_abi.JM_LANE_SOLVER_FAILURE
        # (a state that turned non-finite in the last integrator step of the launch is flagged by the kernel at the NEXT
        # launch: look at it directly, so that the lane restarts now and no NaN reaches the observation)
        # (the acceleration of the end state stands for the state: a non-finite q or v makes it non-finite; one column sum)
        finite = torch.isfinite(self.engine.robot_state.a.sum(0) + self._pipeline_sum())
        truncated = (status != 0) | ~finite | (self._lane_time() >= self.simulation_duration_max)
        return torch.zeros_like(truncated), truncated

    def _pipeline_sum(self) -> Any:
        """Per-lane sum of the controller / observer state (hook of the pipeline environments): non-finite where that
        state is."""
        return 0.0

    def compute_reward(self, terminated: torch.Tensor) -> torch.Tensor:
        return torch.zeros(self.num_envs, dtype=self.dtype, device=self.device)

    def _on_reset(self, lane_mask: Optional[torch.Tensor]) -> None:
        """Hook for controller/observer state (called with the mask of the lanes being reset)."""

    # ------------------------------------------------------------------ gym surface
    def _lane_time(self) -> torch.Tensor:
        return self._clock - self._t0

    def observation(self) -> ObsType:
        rs = self.engine.robot_state
        return {
            "t": self._lane_time(),
            "states": {"agent": {"q": rs.q.T, "v": rs.v.T}},
            "measurements": {k: v.permute(2, 0, 1) for k, v in self.engine.sensor_measurements.items()},
        }

    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None
              ) -> Tuple[ObsType, Dict[str, Any]]:
        """≙ `BaseJiminyEnv.reset(seed, options)` for the whole batch."""
        if seed is not None:
            self._generator.manual_seed(int(seed))
        # the device-side stream (terrain-patch offsets, spawn heights, masked re-draws) is re-seeded from the host
        # generator at its next use: `reset(seed=s)` reproduces all of them, whatever the disturbance settings
        self._dev_gen = None
        # cached reset states / PD targets of a previous graph capture do not outlive the episode batch
        self._state_cache = None
        self._target_cache = None
        self.engine.stop()
        q, v = self._sample_state(self.num_envs)
        self._q0, self._v0 = q, v
        self.engine.field("command").zero_()
        self._on_reset(None)
        if self._ground_profile is not None:
            self.engine.set_ground_profile(*self._ground_profile)
        self._randomise_ground(None)
        self._randomise_flexibility(None)
        if self._spawn_dz is not None:
            q = q.clone()
            q[2] += self._spawn_dz.to(q.dtype).to(q.device)
        self._randomise_sensors()
        self._schedule_disturbances()
        if self._model_options:
            self.engine.seed_model(int(torch.randint(0, 2 ** 31 - 1, (1,), generator=self._generator)))
        self.engine.start(q, v)
        self.num_steps.zero_()
        self._t0.zero_()
        self._clock.zero_()
        return self.observation(), {}

    def step(self, action: torch.Tensor
             ) -> Tuple[ObsType, torch.Tensor, torch.Tensor, torch.Tensor, Dict[str, Any]]:
        """≙ `BaseJiminyEnv.step(action)` (generic.py:761-880)."""
        if not self.engine.is_simulation_running:
            raise RuntimeError("No simulation running. Please call `reset` before `step`.")
        if tuple(action.shape) != (self.num_envs, self.model.nmotors):
            raise ValueError(f"action must have shape ({self.num_envs}, {self.model.nmotors})")
        self._step_engine(action)
        self._clock.fill_(float(self.engine._t))
        reward, terminated, truncated, done = self._after_step()
        info: Dict[str, Any] = {}
        if self.auto_reset and bool(done.any()):
            # gymnasium "next-step" autoreset would cost a launch per step; lanes are reset in
            # place here and the final observation of the finished lanes is not returned
            info["reset_mask"] = done
            self.reset_lanes(done)
        return self.observation(), reward, terminated, truncated, info

    def _after_step(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """Episode bookkeeping after the engine advanced (tensor programs only, no host read-back)."""
        self.num_steps += 1
        terminated, truncated = self.has_terminated()
        reward = torch.nan_to_num(self.compute_reward(terminated), nan=0.0, posinf=0.0, neginf=0.0)   # (numerically failed lanes)
        return reward, terminated, truncated, terminated | truncated

    def _step_engine(self, action: torch.Tensor) -> None:
        self.engine.set_command(self.compute_command(action))
        self._refill_impulses(self.step_dt)
        self.engine.step(self.step_dt)

    def _refill_impulses(self, horizon: float) -> None:
        """Keep the pushes that start within the next `horizon` seconds of engine time registered (see
        `_schedule_disturbances`); a push is drawn once, when it enters the horizon."""
        if getattr(self, "_impulse_frame", None) is None:
            return
        scale = float(self.std_ratio.get("disturbance", 0.0))
        g, B = self._generator, self.num_envs
        t_now = float(self.engine._t)    # (host-side clock: no device synchronisation)
        while True:
            t_ref = self._impulse_index * self.F_IMPULSE_PERIOD
            if t_ref - self.F_IMPULSE_DELTA > t_now + horizon + self.F_IMPULSE_DT:
                break
            t = t_ref + self.F_IMPULSE_DELTA * float(torch.rand(1, generator=g) * 2.0 - 1.0)
            t = max(t, t_now, self._impulse_end)
            d = torch.randn(2, B, generator=g, dtype=torch.float64)
            d = d / d.norm(dim=0, keepdim=True)
            mag = torch.rand(B, generator=g, dtype=torch.float64) * scale * self.F_IMPULSE_SCALE
            f = torch.zeros(6, B, dtype=torch.float64)
            f[:2] = d * mag
            f = f.to(self.device)
            # environments whose episode is younger than the first push of the reference's schedule are spared
            young = (t - self._t0) < (self.F_IMPULSE_PERIOD - self.F_IMPULSE_DELTA)
            f = torch.where(young[None, :], torch.zeros_like(f), f)
            self.engine._schedule_impulse_force(self._impulse_frame, t, self.F_IMPULSE_DT, f)
            self._impulse_end = t + self.F_IMPULSE_DT
            self._impulse_index += 1

    def reset_lanes(self, lane_mask: torch.Tensor) -> None:
        q, v = self._state_cache if self._state_cache is not None else self._sample_state(self.num_envs)
        self._on_reset(lane_mask)
        self._randomise_ground(lane_mask)
        self._randomise_flexibility(lane_mask)
        if self._spawn_dz is not None:
            q = q.clone()
            q[2] += self._spawn_dz.to(q.dtype).to(q.device)
        # a fresh episode starts from a zero command like `reset()` (for the PD pipeline it IS the controller's
        # output at the reset state: target = measured position, zero velocity), not from the last command of the
        # finished episode.  With sensor noise / delay configured the first observation of the re-initialised lanes
        # is the raw measurement (their generators and histories restart at the next sensor refresh).
        cmd = self.engine.field("command")
        cmd.copy_(torch.where(lane_mask[None, :], torch.zeros_like(cmd), cmd))
        if self._model_options and "model_lane" in self.engine._fields:
            self.engine.sample_model_biases(lane_mask)     # a new biased model for the new episode (Model::reset)
        if self._f_xy_profile is not None and float(self.std_ratio.get("disturbance", 0.0)) > 0.0:
            for proc in self._f_xy_profile:
                # `func.reset(self.np_random)` of the new episode; drawn by the device generator (all lanes, the masked
                # ones keep the draw): no host normals, no host -> device copy on the auto-reset path
                proc.reset(self._device_generator(), lane_mask=lane_mask)
        self.engine.reset_lanes(lane_mask, q, v)
        self.num_steps.masked_fill_(lane_mask, 0)
        self._t0.copy_(torch.where(lane_mask, self._clock.expand_as(self._t0), self._t0))

    def _device_generator(self) -> torch.Generator:
        g = getattr(self, "_dev_gen", None)
        if g is None:
            g = self._dev_gen = torch.Generator(device=self.device)
            g.manual_seed(int(torch.randint(0, 2 ** 31 - 1, (1,), generator=self._generator)))
        return g

    GROUND_FOOTPRINT_RADIUS = 0.6
    _spawn_dz: Optional[torch.Tensor] = None

    def _randomise_ground(self, lane_mask: Optional[torch.Tensor]) -> None:
        """Ground friction of the environments being reset, ≙ `sample(*GROUND_FRICTION_RANGE,
        scale=std_ratio['ground'], enable_log_scale=True)` (envs/locomotion.py:28, 257-262 with
        utils/misc.py:178-219: 10 ** (mean + dev * U(-1, 1)), mean = 1.1, dev = 0.9 * scale)."""
        if self._ground_patch_extent is not None and self._ground_profile is not None:
            ext = torch.tensor(self._ground_patch_extent, dtype=torch.float64, device=self.device)[:, None]
            off = (torch.rand((2, self.num_envs), generator=self._device_generator(), dtype=torch.float64, device=self.device) * 2.0 - 1.0) * ext
            if lane_mask is not None and "ground_offset" in self.engine._fields:
                off = torch.where(lane_mask[None, :], off, self.engine.field("ground_offset").to(torch.float64))
            self.engine.set_ground_offsets(off)
            # put the robots down ON their patch: base height raised by the highest ground under the footprint
            self._spawn_dz = self.engine.ground_height_around(self._q0[0:2].to(self.device) + off, self.GROUND_FOOTPRINT_RADIUS)
        scale = float(self.std_ratio.get("ground", 0.0))
        if scale <= 0.0:
            return
        lo, hi = 0.2, 2.0
        u = torch.rand(self.num_envs, generator=self._generator, dtype=torch.float64) * 2.0 - 1.0
        mu = (10.0 ** (0.5 * (lo + hi) + 0.5 * scale * (hi - lo) * u)).to(self.dtype).to(self.device)
        if lane_mask is not None and "friction" in self.engine._fields:
            mu = torch.where(lane_mask, mu, self.engine.field("friction")[0])
        self.engine.set_lane_friction(mu)

    FLEX_STIFFNESS_SCALE, FLEX_DAMPING_SCALE = 1000.0, 10.0    # envs/locomotion.py:37-38

    def _randomise_flexibility(self, lane_mask: Optional[torch.Tensor]) -> None:
        """Flexibility parameters of the environments being reset, ≙ `flexibility['stiffness'] += FLEX_STIFFNESS_SCALE *
        sample(scale=std_ratio['model'])`, `flexibility['damping'] += FLEX_DAMPING_SCALE * sample(...)` for every entry of
        `flexibilityConfig` (envs/locomotion.py:288-296; utils/misc.py:178-212: `sample(scale=s)` = U(-s, s)): one draw per
        flexibility joint and environment, added to the three axes alike, around the model's own values (the reference
        starts every episode from the robot options it was built with).  Values are kept non-negative."""
        scale = float(self.std_ratio.get("model", 0.0))
        flex = self.model.flexibility_joint_indices
        if scale <= 0.0 or not flex or self.model.flex_stiffness is None:
            return
        n, B = len(flex), self.num_envs
        k0 = torch.tensor(np.asarray(self.model.flex_stiffness)[flex], dtype=torch.float64)[:, :, None]
        d0 = torch.tensor(np.asarray(self.model.flex_damping)[flex], dtype=torch.float64)[:, :, None]
        u = torch.rand((2, n, 1, B), generator=self._generator, dtype=torch.float64) * 2.0 - 1.0
        k = (k0 + self.FLEX_STIFFNESS_SCALE * scale * u[0]).clamp_min(0.0).to(self.dtype).to(self.device)
        d = (d0 + self.FLEX_DAMPING_SCALE * scale * u[1]).clamp_min(0.0).to(self.dtype).to(self.device)
        if lane_mask is not None and "flexibility" in self.engine._fields:
            old = self.engine.field("flexibility").reshape(n, 2, 3, B)
            k = torch.where(lane_mask[None, None, :], k, old[:, 0])
            d = torch.where(lane_mask[None, None, :], d, old[:, 1])
        self.engine.set_lane_flexibility(k, d)

    # envs/locomotion.py:30-36
    F_IMPULSE_DT, F_IMPULSE_PERIOD, F_IMPULSE_DELTA, F_IMPULSE_SCALE = 10.0e-3, 2.0, 0.25, 1000.0
    F_PROFILE_SCALE, F_PROFILE_WAVELENGTH, F_PROFILE_PERIOD = 50.0, 0.2, 1.0
    _f_xy_profile = None

    def _schedule_disturbances(self) -> None:
        """≙ the disturbance part of `WalkerJiminyEnv._setup` (envs/locomotion.py:298-326): every F_IMPULSE_PERIOD
        seconds (+- F_IMPULSE_DELTA) a horizontal push of random direction and magnitude
        U(0, std_ratio['disturbance'] * F_IMPULSE_SCALE) on the root body, one draw per environment, plus the
        continuous Gaussian-process force of :327-359 (below).

        The reference draws the pushes of a whole episode in episode time at every `_setup`.  Here the environments
        of a batch share the engine clock and its breakpoints while their episodes start at different times
        (auto-reset), so the schedule is periodic in ENGINE time and generated lazily -- only the next push is
        registered (`_refill_impulses`, called before every engine step), whatever `simulation_duration_max` -- and an
        environment is spared by the pushes of the first F_IMPULSE_PERIOD - F_IMPULSE_DELTA seconds of its own
        episode, in which the reference schedules none."""
        scale = float(self.std_ratio.get("disturbance", 0.0))
        self.engine.remove_all_forces()
        self._impulse_frame = None
        if scale <= 0.0 or not self.model.has_freeflyer:
            return
        frame = next(n for n, f in self.model.frames.items() if f.parent_joint == 1)
        g = self._generator
        B = self.num_envs
        self.engine._force_frame_index(frame)    # binds the `applied` field while no simulation is running
        self._impulse_frame = frame
        self._impulse_index = 1                  # the push of period k starts at k * F_IMPULSE_PERIOD +- delta
        self._impulse_end = 0.0
        # the continuous part (envs/locomotion.py:165-167, 327-359): two periodic Gaussian processes, one realisation
        # per environment, drive the x / y force on the root body: F_PROFILE_SCALE * std_ratio * process(episode time)
        from .processes import PeriodicGaussianProcess
        if self._f_xy_profile is None:
            self._f_xy_profile = [PeriodicGaussianProcess(self.F_PROFILE_WAVELENGTH, self.F_PROFILE_PERIOD, B, self.dtype, self.device),
                                  PeriodicGaussianProcess(self.F_PROFILE_PERIOD, self.F_PROFILE_PERIOD, B, self.dtype, self.device)]
        self._dev_gen = None          # re-seeded from the (just re-seeded) host generator at its next use
        for proc in self._f_xy_profile:
            proc.reset(g)

        def profile(t: float, q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
            w = torch.zeros((6, B), dtype=self.dtype, device=self.device)
            tl = t - self._t0
            w[0] = self.F_PROFILE_SCALE * scale * self._f_xy_profile[0](tl)
            w[1] = self.F_PROFILE_SCALE * scale * self._f_xy_profile[1](tl)
            return w
        self.engine.register_profile_force(frame, profile)

    # scales of envs/locomotion.py:40-61 (delay [s]; noise and bias per field)
    SENSOR_DELAY_SCALE = {"EncoderSensor": 3.0e-3, "EffortSensor": 0.0, "ContactSensor": 0.0, "ForceSensor": 0.0,
                          "ImuSensor": 0.0}
    SENSOR_NOISE_SCALE = {"EncoderSensor": (0.0, 0.02), "EffortSensor": (10.0,), "ContactSensor": (2.0, 2.0, 2.0),
                          "ForceSensor": (2.0, 2.0, 2.0, 10.0, 10.0, 10.0),
                          "ImuSensor": (0.0, 0.0, 0.0, 0.01, 0.01, 0.01, 0.2, 0.2, 0.2)}

    def _randomise_sensors(self) -> None:
        """Sensor noise, bias, delay and jitter, ≙ envs/locomotion.py:264-288: for every sensor
        `delay, jitter ~ U(0, s * SENSOR_DELAY_SCALE)`, `bias, noiseStd ~ s * SENSOR_NOISE_SCALE * U(-1, 1)`
        (the reference scales both with the *noise* table; a negative standard deviation is its own business:
        the magnitude is used here).  Called by `reset()` while no simulation is running."""
        scale = float(self.std_ratio.get("sensors", 0.0))
        if scale <= 0.0:
            return
        rg = np.random.default_rng(int(torch.randint(0, 2 ** 31 - 1, (1,), generator=self._generator)))
        any_rng = False
        for stype, table in self.SENSOR_NOISE_SCALE.items():
            n = len(self.model.sensors.get(stype, []))
            if n == 0:
                continue
            sc = scale * np.asarray(table)
            nf = self.engine._SENSOR_FIELDS[stype][1]
            nb = 9 if stype == "ImuSensor" else nf
            bias = rg.uniform(-1.0, 1.0, (n, len(sc))) * sc
            std = np.abs(rg.uniform(-1.0, 1.0, (n, len(sc))) * sc)
            if stype == "ImuSensor":
                # 9 bias entries (rotation, gyro, accel) but 6 measured fields: noise on the last 6
                std = std[:, 3:]
            else:
                bias, std = bias[:, :nb], std[:, :nf]
            dmax = scale * self.SENSOR_DELAY_SCALE[stype]
            delay, jitter = rg.uniform(0.0, dmax, n), rg.uniform(0.0, dmax, n)
            self.engine.set_sensor_options(stype, noise_std=std, bias=bias, delay=delay, jitter=jitter)
            any_rng = True
        if any_rng:
            self.engine.seed_sensors(int(rg.integers(0, 2 ** 31 - 1)))

    def close(self) -> None:
        self.engine.stop()


class WalkerVecEnv(VecJiminyEnv):
    """≙ `WalkerJiminyEnv` (envs/locomotion.py): legged robot with a free-flyer, fall detection
    at half the neutral height and the 'survival' / 'energy' / 'failure' / 'direction' reward mixture.
    `std_ratio['model']` randomises the flexibility stiffness / damping of every environment per episode
    (locomotion.py:288-296, `VecJiminyEnv._randomise_flexibility`)."""

    def __init__(self, *args: Any, reward_mixture: Optional[Dict[str, float]] = None, **kw: Any) -> None:
        super().__init__(*args, **kw)
        self.reward_mixture = dict(reward_mixture or {"survival": 1.0})
        m = self.model
        lim = torch.tensor([mo.effort_limit * mo.velocity_limit for mo in m.motors], dtype=self.dtype)
        self._power_consumption_max = float(lim.sum())  # locomotion.py:230-236
        # encoder of every motor (the reference indexes the encoder data through `encoder_to_motor_map`)
        enc_of = {e.get("motor_index", -1): i for i, e in enumerate(m.sensors.get("EncoderSensor", []))}
        self._motor_enc_idx = (torch.tensor([enc_of[i] for i in range(m.nmotors)], device=self.device)
                               if all(i in enc_of for i in range(m.nmotors)) and m.nmotors else None)

        # 'direction' (locomotion.py:419-424): mean of the logged free-flyer Y position over the episode, per environment.
        # The reference's log holds one row per integrator step; here the position is sampled once per environment step
        # (what a launch makes visible) plus the initial state, accumulated on the device.
        self._dir_sum = torch.zeros(self.num_envs, dtype=torch.float64, device=self.device)
        self._dir_n = torch.zeros(self.num_envs, dtype=torch.float64, device=self.device)

    def _direction_restart(self, lane_mask: Optional[torch.Tensor]) -> None:
        y0 = self.engine.robot_state.q[1].to(torch.float64)
        if lane_mask is None:
            self._dir_sum.copy_(y0)
            self._dir_n.fill_(1.0)
        else:
            self._dir_sum.copy_(torch.where(lane_mask, y0, self._dir_sum))
            self._dir_n.copy_(torch.where(lane_mask, torch.ones_like(self._dir_n), self._dir_n))

    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        out = super().reset(seed, options)
        self._direction_restart(None)
        return out

    def reset_lanes(self, lane_mask: torch.Tensor) -> None:
        super().reset_lanes(lane_mask)
        self._direction_restart(lane_mask)

    def _after_step(self):
        self._dir_sum += self.engine.robot_state.q[1].to(torch.float64)
        self._dir_n += 1.0
        return super()._after_step()

    def has_terminated(self) -> Tuple[torch.Tensor, torch.Tensor]:
        terminated, truncated = super().has_terminated()
        z = self.engine.robot_state.q[2]
        return terminated | (z < 0.5 * self._height_neutral), truncated  # locomotion.py:381-383

    def compute_reward(self, terminated: torch.Tensor) -> torch.Tensor:
        total = torch.zeros(self.num_envs, dtype=self.dtype, device=self.device)
        if "survival" in self.reward_mixture:
            total += self.reward_mixture["survival"]
        if "energy" in self.reward_mixture:
            if self._motor_enc_idx is None:
                raise RuntimeError("the 'energy' reward needs one motor-side encoder per motor")
            enc = self.engine.sensor_measurements["EncoderSensor"]  # (2, n_enc, B), encoder order
            power = torch.clamp_min(self.engine.command * enc[1][self._motor_enc_idx], 0.0).sum(0)
            total -= self.reward_mixture["energy"] * power / self._power_consumption_max
        if "failure" in self.reward_mixture:
            total -= self.reward_mixture["failure"] * terminated.to(self.dtype)
        if "direction" in self.reward_mixture:
            # at termination only: minus the absolute mean lateral position of the episode (locomotion.py:419-424)
            drift = (self._dir_sum / torch.clamp_min(self._dir_n, 1.0)).abs().to(self.dtype)
            total -= self.reward_mixture["direction"] * drift * terminated.to(self.dtype)
        return total


class PDControlledWalkerVecEnv(WalkerVecEnv):
    """Walker with the reference's low-level pipeline on device (gym_jiminy/envs/anymal.py:82-127):
    `PDAdapter` (order 1: the action is the target motor velocity) -> `PDController` (ZOH command
    integrator + PD law, run at the controller period) and a `MahonyFilter` observer per IMU.

    The engine advances one controller period per launch; the command is refreshed between
    launches from the encoder rows written by the previous launch, like the reference's
    `_controller_handle` called from `Engine::step` at each controller breakpoint.
    """

    def __init__(self, model: CompiledModel, num_envs: int, step_dt: float, control_dt: float,
                 kp: Any, kd: Any, mahony_kp: float = 1.0, mahony_ki: float = 0.1,
                 joint_position_margin: float = 0.0, joint_velocity_limit: float = float("inf"),
                 joint_acceleration_limit: Optional[float] = None,
                 safety_limit: Optional[Dict[str, float]] = None, **kw: Any) -> None:
        opts = kw.pop("engine_options", None) or {}
        st = dict(opts.get("stepper", {}))
        st.setdefault("controllerUpdatePeriod", control_dt)
        st.setdefault("sensorsUpdatePeriod", control_dt)
        opts = dict(opts, stepper=st)
        super().__init__(model, num_envs, step_dt, engine_options=opts, **kw)
        self.control_dt = float(control_dt)
        self._n_ctrl = int(round(step_dt / control_dt))
        if abs(self._n_ctrl * control_dt - step_dt) > 1e-9:
            raise ValueError("step_dt must be a multiple of the controller period")
        M, B, dev, dt_ = model.nmotors, self.num_envs, self.device, self.dtype
        # encoders in motor order
        enc_of = {e["motor_index"]: i for i, e in enumerate(model.sensors["EncoderSensor"])
                  if e["motor_index"] >= 0}
        if sorted(enc_of) != list(range(M)):
            raise ValueError("the PD pipeline needs one motor-side encoder per motor")
        self._enc_idx = torch.tensor([enc_of[i] for i in range(M)], device=dev)
        self.kp = torch.as_tensor(kp, dtype=dt_, device=dev)
        self.kd = torch.as_tensor(kd, dtype=dt_, device=dev)
        self.effort_limit = torch.tensor([m.effort_limit for m in model.motors], dtype=dt_, device=dev)
        # command-state bounds of `PDController.__init__` (proportional_derivative_controller.py:405-437):
        # motor-side position limits shrunk by the margin, velocity min(motor limit, reduction * joint limit),
        # acceleration reduction * joint limit -- or, without one, the largest that still allows bang-bang control
        red = np.array([m.reduction for m in model.motors])
        p_lo = np.array([model.position_lower[m.idx_q] * m.reduction for m in model.motors]) + red * joint_position_margin
        p_hi = np.array([model.position_upper[m.idx_q] * m.reduction for m in model.motors]) - red * joint_position_margin
        v_lim = np.minimum(np.array([m.velocity_limit for m in model.motors]), red * joint_velocity_limit)
        if joint_acceleration_limit is None:
            kp_, kd_ = np.broadcast_to(np.asarray(kp, dtype=float), (M,)), np.broadcast_to(np.asarray(kd, dtype=float), (M,))
            eff = np.array([m.effort_limit for m in model.motors])
            a_lim = np.minimum(2.0 * v_lim / step_dt, eff / (kp_ * step_dt * np.maximum(step_dt, kd_)))
        else:
            a_lim = red * float(joint_acceleration_limit)
        lo = torch.tensor(np.stack([p_lo, -v_lim, -a_lim]), dtype=dt_, device=dev)
        hi = torch.tensor(np.stack([p_hi, v_lim, a_lim]), dtype=dt_, device=dev)
        self.command_state_lower, self.command_state_upper = lo, hi
        self.command_state = torch.zeros((3, M, B), dtype=dt_, device=dev)
        self._accel = torch.zeros((M, B), dtype=dt_, device=dev)
        self._torque = torch.zeros((M, B), dtype=dt_, device=dev)
        n_imu = len(model.sensors["ImuSensor"])
        self.mahony_kp, self.mahony_ki = float(mahony_kp), float(mahony_ki)
        self.imu_quat = torch.zeros((4, n_imu, B), dtype=dt_, device=dev)
        self.imu_quat[3] = 1.0
        self._bias = torch.zeros((3, n_imu, B), dtype=dt_, device=dev)
        self._omega = torch.zeros_like(self._bias)
        self._cf = torch.zeros_like(self._bias)
        # per-tick blocks as single HIP launches (JIMINY_AMD_TENSOR_BLOCKS=1 selects the tensor
        # programs of blocks.py instead: A/B measurements and debugging only)
        import os
        self._hip_blocks = None
        if os.environ.get("JIMINY_AMD_TENSOR_BLOCKS", "0") != "1":
            self._hip_blocks = blocks.HipBlocks(self.engine, self._enc_idx, lo, hi, self.kp, self.kd,
                                                self.effort_limit)
        # optional `MotorSafetyLimit` between the PD controller and the motors (gym_jiminy blocks/motor_safety_limit.py:
        # 81-175; Atlas: kp = 1 / MOTOR_POSITION_MARGIN, kd = MOTOR_VELOCITY_SAFE_GAIN): motor-side soft bounds
        self._safety = None
        if safety_limit is not None:
            if self._hip_blocks is None:
                raise NotImplementedError("the motor safety limit runs as a HIP block")
            margin = float(safety_limit.get("soft_position_margin", 0.0))
            vmax = float(safety_limit["soft_velocity_max"])
            if margin < 0.0 or vmax < 0.0:
                raise ValueError("Soft position margin and soft maximum velocity must be positive.")
            self._safety = dict(
                kp=np.full(M, float(safety_limit["kp"])), kd=np.full(M, float(safety_limit["kd"])),
                lo=np.array([model.position_lower[m.idx_q] * m.reduction for m in model.motors]) + red * margin,
                hi=np.array([model.position_upper[m.idx_q] * m.reduction for m in model.motors]) - red * margin,
                vlim=np.minimum(np.array([m.velocity_limit for m in model.motors]), red * vmax))

    def _encoders(self) -> torch.Tensor:
        enc = self.engine.sensor_measurements["EncoderSensor"]   # (2, n_enc, B)
        return enc[:, self._enc_idx]

    def _pipeline_sum(self) -> Any:
        # (an infinite IMU sample of a lane that is about to fail poisons its attitude estimate one step before its state)
        return self.imu_quat.sum((0, 1))

    def _on_reset(self, lane_mask: Optional[torch.Tensor]) -> None:
        q0, _ = self._state_cache if self._state_cache is not None else self._sample_state(self.num_envs)
        if lane_mask is None or getattr(self, "_target_cache", None) is None:
            self._target_cache = torch.stack([q0[m.idx_q] * m.reduction for m in self.model.motors])
        target = self._target_cache
        if lane_mask is None:
            self._graph = None      # a full reset restarts the engine: the step graph is captured again
            self.command_state.zero_()
            self.command_state[0] = target
            self.imu_quat.zero_(); self.imu_quat[3] = 1.0
            self._bias.zero_()
        else:
            # by selection, never by arithmetic: a lane truncated for JM_LANE_NAN carries NaN in its controller /
            # observer state, and NaN * 0 stays NaN
            m = lane_mask[None, None, :]
            fresh = torch.zeros_like(self.command_state)
            fresh[0] = target
            self.command_state.copy_(torch.where(m, fresh, self.command_state))
            unit = torch.zeros_like(self.imu_quat)
            unit[3] = 1.0
            self.imu_quat.copy_(torch.where(m, unit, self.imu_quat))
            for t in (self._bias, self._omega, self._cf):
                t.copy_(torch.where(m, torch.zeros_like(t), t))
            for t in (self._accel, self._torque):
                t.copy_(torch.where(lane_mask[None, :], torch.zeros_like(t), t))

    # ------------------------------------------------------------------ HIP-graph replay of one environment step
    def enable_graph(self, enable: bool = True, whole_step: bool = False) -> None:
        """Replay the launches of one environment step -- PD adapter, then per controller tick PD controller -> physics
        launch -> Mahony filter: 26 launches for ANYmal -- as ONE captured HIP graph instead of issuing them from
        Python.  For small batches per GPU (a sharded config 4 / 5: a few thousand environments) the step is bound by
        the host's launch rate, not by the kernels; at B = 65 536 it makes no difference.  Needs the HIP blocks, a
        fixed-step solver, no sensor noise / delay and no applied forces (their host-side schedules run between the
        launches).  The graph is captured at the next `step` and dropped by `reset`.

        `whole_step=True` captures the WHOLE `env.step`: the physics chain, the episode clock, termination / truncation,
        the reward and an unconditional masked auto-reset (`reset_lanes` with the `done` mask: lanes that are not done
        leave its launches at once) -- no host read-back at all (`bool(done.any())` of the eager step is the one
        synchronisation per environment step).  `step` then returns static tensors (`reward`, `terminated`, `truncated`,
        `info["reset_mask"]`) that the next `step` overwrites.  Additionally needs `auto_reset`, no ground-friction
        randomisation and the deterministic default state sampler (model biases are fine: drawn on the device)."""
        if enable:
            eng = self.engine
            if self._hip_blocks is None or eng._adaptive is not None or eng._sensor_noise or eng._impulse_forces or eng._profile_forces \
                    or getattr(self, "_impulse_frame", None) is not None:
                raise NotImplementedError("enable_graph needs the HIP blocks, a fixed-step solver, noiseless sensors and "
                                          "no applied forces")
            if whole_step and (not self.auto_reset or float(self.std_ratio.get("ground", 0.0)) > 0.0 or
                               self._ground_patch_extent is not None):
                raise NotImplementedError("whole-step graphs need auto_reset and no ground-friction / terrain-patch randomisation")
        self._graph_enabled = bool(enable)
        self._graph_whole = bool(enable and whole_step)
        self._graph = None
        self._state_cache = None
        self._target_cache = None

    def _graph_preconditions(self) -> None:
        """Checked again when the graph is captured: `reset()` may have registered disturbance forces (or the user sensor
        noise) after `enable_graph` was called, and their host-side schedules do not run inside a replay."""
        eng = self.engine
        if eng._adaptive is not None or eng._sensor_noise or eng._impulse_forces or eng._profile_forces \
                or getattr(self, "_impulse_frame", None) is not None:
            raise NotImplementedError("enable_graph needs a fixed-step solver, noiseless sensors and no applied forces "
                                      "(std_ratio['disturbance'] registers forces at reset())")
        if self._graph_whole:
            a, b = self._sample_state(self.num_envs), self._sample_state(self.num_envs)
            if not (torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])):
                raise NotImplementedError("whole-step graphs need a deterministic reset-state sampler")

    def _step_engine_graphed(self, action: torch.Tensor) -> None:
        eng = self.engine
        if self._graph is None:
            self._graph_preconditions()
            # the launches carry no time: the breakpoint plan of a step must be the same at every step
            plan = _plan_signature(eng, self.control_dt, self._n_ctrl)
            self._g_action = torch.zeros((self.model.nmotors, self.num_envs), dtype=self.dtype, device=self.device)
            host = (eng._t, eng._t_prev, eng._t_error, eng._iter, eng._dt, eng._command_dirty)
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(self.device)
            if self._graph_whole:
                self._state_cache = self._sample_state(self.num_envs)    # (host -> device copies are not capturable)
            with torch.cuda.graph(g):
                self._issue_step(self._g_action)          # recorded, not executed
                if self._graph_whole:
                    reward, terminated, truncated, done = self._after_step()
                    self._g_out = (reward, terminated, truncated, done)
                    self.reset_lanes(done)
            after = (eng._t, eng._iter, eng._dt)
            eng._t, eng._t_prev, eng._t_error, eng._iter, eng._dt, eng._command_dirty = host
            self._graph = (g, plan, after[0] - host[0], after[1] - host[3], after[2])
            self._graph_replays = 0
        g, plan, dt_total, d_iter, dt_last = self._graph
        self._graph_replays += 1
        if self._graph_replays % 64 == 0 and _plan_signature(eng, self.control_dt, self._n_ctrl) != plan:
            self._graph = None                            # (never seen: the plan is periodic; re-capture if it is not)
            return self._step_engine_graphed(action)
        self._g_action.copy_(action.to(self.dtype).T)
        # host-side bookkeeping of the `n_ctrl` engine steps the graph stands for (Kahan-compensated like engine.step);
        # the device clock gets the end time before the replay: the captured episode bookkeeping reads it
        for _ in range(self._n_ctrl):
            corrected = self.control_dt - eng._t_error
            t_end = eng._t + corrected
            eng._t_error = (t_end - eng._t) - corrected
            eng._t_prev, eng._t = eng._t, t_end
        self._clock.fill_(float(eng._t))
        g.replay()
        eng._iter += d_iter
        eng._dt = dt_last
        eng._command_dirty = False

    def step(self, action: torch.Tensor):
        # (the first step of a simulation carries the reference's opening microsecond step -- `engine.substep_sizes` --:
        # it is issued eagerly, the periodic plan is captured from the second step on)
        if not getattr(self, "_graph_whole", False):
            return super().step(action)
        if self.engine._opening_step:
            obs, reward, terminated, truncated, info = super().step(action)
            info.setdefault("reset_mask", terminated | truncated)
            return obs, reward, terminated, truncated, info
        if not self.engine.is_simulation_running:
            raise RuntimeError("No simulation running. Please call `reset` before `step`.")
        if tuple(action.shape) != (self.num_envs, self.model.nmotors):
            raise ValueError(f"action must have shape ({self.num_envs}, {self.model.nmotors})")
        self._step_engine_graphed(action)
        reward, terminated, truncated, done = self._g_out
        return self.observation(), reward, terminated, truncated, {"reset_mask": done}

    def _step_engine(self, action: torch.Tensor) -> None:
        if getattr(self, "_graph_enabled", False) and not self.engine._opening_step:
            self._step_engine_graphed(action)
        else:
            self._refill_impulses(self.step_dt)
            self._issue_step(action.to(self.dtype).T)

    def _issue_step(self, a: torch.Tensor) -> None:
        hb = self._hip_blocks
        if hb is not None:
            hb.pd_adapter(a.contiguous(), 1, self.command_state, False, None, self.step_dt, self._accel)
        else:
            blocks.pd_adapter(a, 1, self.command_state, self.command_state_lower, self.command_state_upper,
                              False, None, self.step_dt, self._accel)
        self.command_state[2].copy_(self._accel)
        for _ in range(self._n_ctrl):
            if hb is not None:
                # torques go straight into the engine's command rows
                hb.pd_controller(self.command_state, self.control_dt, self.engine.field("command"))
                if self._safety is not None:
                    sf = self._safety
                    cmd = self.engine.field("command")
                    hb.motor_safety_limit(cmd, sf["kp"], sf["kd"], sf["lo"], sf["hi"], sf["vlim"], cmd)
                self.engine.mark_command_changed()
            else:
                blocks.pd_controller(self._encoders(), self.command_state, self.command_state_lower,
                                     self.command_state_upper, self.kp, self.kd, self.effort_limit,
                                     self.control_dt, self._torque)
                self.engine.set_command(self._torque)
            self.engine.step(self.control_dt)
            if hb is not None:
                hb.mahony_filter(self.imu_quat, self._omega, self._cf, self._bias,
                                 self.mahony_kp, self.mahony_ki, self.control_dt)
            else:
                imu = self.engine.sensor_measurements["ImuSensor"]     # (6, n_imu, B)
                blocks.mahony_filter(self.imu_quat, self._omega, self._cf, imu[:3], imu[3:], self._bias,
                                     self.mahony_kp, self.mahony_ki, self.control_dt)

    def observation(self) -> ObsType:
        obs = super().observation()
        obs["features"] = {"mahony_filter": self.imu_quat.permute(2, 0, 1)}
        obs["actions"] = {"pd_controller": self.command_state[:2].permute(2, 0, 1)}
        return obs


def _plan_signature(eng: Any, control_dt: float, n_ctrl: int) -> tuple:
    """The launches `n_ctrl` engine steps of `control_dt` would issue from the engine's current time."""
    from .engine import plan_step
    t, t_err, out = eng._t, eng._t_error, []
    for _ in range(n_ctrl):
        launches, t, t_err = plan_step(t, t_err, control_dt, eng._options)
        out.append(tuple((round(dt / 1e-12), n, bool(c), bool(sn)) for dt, n, c, sn in launches))
    return tuple(out)


# constants of the reference ANYmal environment (python/gym_jiminy/envs/gym_jiminy/envs/anymal.py:17-35)
ANYMAL_STEP_DT = 0.04
ANYMAL_CONTROL_DT = 0.005
ANYMAL_PD_KP = (1500.0,) * 12
ANYMAL_PD_KD = (0.01,) * 12
ANYMAL_MAHONY_KP, ANYMAL_MAHONY_KI = 1.0, 0.1
ANYMAL_MOTOR_VELOCITY_MAX = 4.0
ANYMAL_MOTOR_ACCELERATION_MAX = 30.0
ANYMAL_SIMULATION_DURATION = 20.0


def make_anymal_env(num_envs: int, dtype: torch.dtype = torch.float64,
                    device: Optional[torch.device] = None, pd_pipeline: bool = True,
                    ode_solver: str = "euler_explicit", dt_max: float = 1e-3,
                    contact_model: str = "spring_damper", **kw: Any) -> VecJiminyEnv:
    """ANYmal with the reference's env constants. `contact_model="constraint"` selects what the
    shipped option file selects (anymal_options.toml:24: joint bounds and contact points as
    constraints, PGS); the default stays the spring-damper model of the north-star configuration."""
    model = load_builtin("anymal")
    opts = {"stepper": {"odeSolver": ode_solver, "dtMax": dt_max,
                        "controllerUpdatePeriod": ANYMAL_CONTROL_DT,
                        "sensorsUpdatePeriod": ANYMAL_CONTROL_DT},
            "contacts": {"model": contact_model}}
    kw.setdefault("simulation_duration_max", ANYMAL_SIMULATION_DURATION)
    if pd_pipeline:
        return PDControlledWalkerVecEnv(model, num_envs, ANYMAL_STEP_DT, ANYMAL_CONTROL_DT,
                                        ANYMAL_PD_KP, ANYMAL_PD_KD, ANYMAL_MAHONY_KP, ANYMAL_MAHONY_KI,
                                        joint_position_margin=0.0,
                                        joint_velocity_limit=ANYMAL_MOTOR_VELOCITY_MAX,
                                        joint_acceleration_limit=ANYMAL_MOTOR_ACCELERATION_MAX,
                                        engine_options=opts, dtype=dtype, device=device, **kw)
    return WalkerVecEnv(model, num_envs, ANYMAL_STEP_DT, engine_options=opts, dtype=dtype,
                        device=device, **kw)


# constants of the reference Atlas environment (python/gym_jiminy/envs/gym_jiminy/envs/atlas.py:24-80, atlas_options.toml)
ATLAS_STEP_DT = 0.04
ATLAS_CONTROL_DT = 0.005
ATLAS_SIMULATION_DURATION = 20.0
ATLAS_NEUTRAL_SAGITTAL_HIP_ANGLE = 0.2
ATLAS_MOTOR_POSITION_MARGIN = 0.02
ATLAS_MOTOR_VELOCITY_SAFE_GAIN = 0.15
ATLAS_MOTOR_VELOCITY_MAX = 4.0
ATLAS_MOTOR_ACCELERATION_MAX = 30.0
ATLAS_PD_REDUCED_KP = (5000.0, 5000.0, 8000.0, 4000.0, 8000.0, 5000.0) * 2      # legs: HpZ, HpX, HpY, KnY, AkY, AkX
ATLAS_PD_REDUCED_KD = (0.01, 0.02, 0.02, 0.01, 0.025, 0.01) * 2
ATLAS_PD_FULL_KP = ((5000.0, 8000.0, 5000.0)                                      # back: Z, Y, X
                    + (500.0, 100.0, 200.0, 500.0, 10.0, 100.0, 10.0)             # left arm
                    + (100.0,)                                                    # neck
                    + (500.0, 100.0, 200.0, 500.0, 10.0, 100.0, 10.0)             # right arm
                    + ATLAS_PD_REDUCED_KP)
ATLAS_PD_FULL_KD = ((0.01, 0.015, 0.02) + (0.01, 0.01, 0.01, 0.02, 0.01, 0.02, 0.02) + (0.01,)
                    + (0.01, 0.01, 0.01, 0.02, 0.01, 0.02, 0.02) + ATLAS_PD_REDUCED_KD)
ATLAS_MAHONY_KP, ATLAS_MAHONY_KI = 0.75, 0.057
# `AtlasJiminyEnv._neutral` (atlas.py:147-164): the arms folded along the body, a slight forward lean of the back
ATLAS_NEUTRAL_JOINTS = {"back_bky": 0.2, "l_arm_elx": 0.2, "l_arm_shx": -math.pi / 2.0, "l_arm_shz": math.pi / 4.0,
                        "l_arm_ely": math.pi / 4.0 + math.pi / 2.0, "r_arm_elx": -0.2, "r_arm_shx": math.pi / 2.0,
                        "r_arm_shz": -math.pi / 4.0, "r_arm_ely": math.pi / 4.0 + math.pi / 2.0}


class AtlasPDControlVecEnv(PDControlledWalkerVecEnv):
    """≙ `AtlasPDControlJiminyEnv` (atlas.py:254-309): MotorSafetyLimit -> PDController -> PDAdapter (order 1) ->
    MahonyFilter around `AtlasJiminyEnv`, whose neutral configuration folds the arms."""

    def _sample_state_numpy(self) -> Tuple[np.ndarray, np.ndarray]:
        m = self.model
        q = m.neutral()
        for name, value in ATLAS_NEUTRAL_JOINTS.items():
            q[int(m.idx_q[m.joint_names.index(name)])] = value
        mask = m.bounded_position_mask()
        q[mask] = np.clip(q[mask], m.position_lower[mask], m.position_upper[mask])
        q[2] -= float(lowest_contact_height(m, q)[0])
        return q, np.zeros(m.nv)


def make_atlas_env(num_envs: int, dtype: torch.dtype = torch.float64, device: Optional[torch.device] = None,
                   ode_solver: str = "euler_explicit", dt_max: float = 1e-3, contact_model: str = "constraint",
                   **kw: Any) -> AtlasPDControlVecEnv:
    """Atlas with the reference's environment constants (atlas.py, atlas_options.toml: explicit Euler at 1 ms, 5 ms
    controller / sensor period, constraint contact model).  All 32 box-vertex contact points of the compiled model are
    kept (the reference environment prunes the ones off the bottom convex hull of each foot, atlas.py:99-115: they never
    touch a flat ground)."""
    model = load_builtin("atlas")
    opts = {"stepper": {"odeSolver": ode_solver, "dtMax": dt_max, "controllerUpdatePeriod": ATLAS_CONTROL_DT,
                        "sensorsUpdatePeriod": ATLAS_CONTROL_DT},
            "contacts": {"model": contact_model}}
    kw.setdefault("simulation_duration_max", ATLAS_SIMULATION_DURATION)
    return AtlasPDControlVecEnv(
        model, num_envs, ATLAS_STEP_DT, ATLAS_CONTROL_DT, ATLAS_PD_FULL_KP, ATLAS_PD_FULL_KD, ATLAS_MAHONY_KP, ATLAS_MAHONY_KI,
        joint_position_margin=0.0, joint_velocity_limit=ATLAS_MOTOR_VELOCITY_MAX,
        joint_acceleration_limit=ATLAS_MOTOR_ACCELERATION_MAX,
        safety_limit={"kp": 1.0 / ATLAS_MOTOR_POSITION_MARGIN, "kd": ATLAS_MOTOR_VELOCITY_SAFE_GAIN,
                      "soft_position_margin": 0.0, "soft_velocity_max": ATLAS_MOTOR_VELOCITY_MAX},
        engine_options=opts, dtype=dtype, device=device, **kw)
