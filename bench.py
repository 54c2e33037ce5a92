#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched hot path on N MI355X (one process per GPU).

A "step" = one pass of the hot path over one batch: every lane advanced by one fixed integrator
step dt = 1e-3 s with the command held, `odeSolver = "runge_kutta_4"` (4 dynamics evaluations:
FK + spring-damper contacts + motor law + ABA), then the extra terms and the sensor refresh
(the unit of work of BASELINE.md section 3).  Workload = BASELINE.json configs[2]: ANYmal
(nq 19, nv 18, 12 motors, 4 contact points), batch 65 536 per GPU (weak scaling), float64 like the
reference, seeded synthetic states resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel (`k_quad` for ANYmal) vs the HBM roof, from per-launch HIP-event timing on the
                launch stream; `traffic` = PMC-measured HBM bytes per launch when profiles/ holds it
  cpu_baseline  the CPU oracle ("port" of the reference's single-threaded algorithm) timed on the
                host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` started by hand (no WORLD_SIZE in the environment): re-exec this very
    command line under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1, and hand its
    exit code back.  (The driver launches the ranks itself; this path makes the script self-contained.)"""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
# float64 vector issue: one wave64 VALU instruction occupies a SIMD for 4 cycles (16 lanes / clk,
# 78.6 TFLOP/s fp64 vector peak = 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz); fp32: 2 cycles
SIMDS, CLOCK_HZ = 1024, 2.4e9


def _abi_rows(model):
    from jiminy_amd import _abi
    return _abi.constraint_rows(model)


def algorithmic_scalars(model) -> int:
    """Scalars that must cross HBM once per env-step (SURVEY.md 8d / BASELINE.md section 3):
    reads q, v, a_prev, command; writes q, v, a; writes the sensor outputs."""
    s = model.sensors
    obs = 6 * len(s.get("ImuSensor", [])) + 6 * len(s.get("ForceSensor", [])) \
        + 3 * len(s.get("ContactSensor", [])) + 2 * len(s.get("EncoderSensor", [])) \
        + len(s.get("EffortSensor", []))
    return (model.nq + 2 * model.nv + model.nmotors) + (model.nq + 2 * model.nv) + obs


def cpu_baseline(model, states, dt: float, budget_s: float = 12.0, solver: str = "runge_kutta_4",
                 constraint_options=None):
    """Oracle timed on the host: 1 thread (the reference's shape: one engine, one robot, one
    thread), then all cores with independent slices (the SubprocVecEnv analogue without IPC)."""
    from oracle.oracle_py import OracleEngine
    from tests.helpers import alloc_constraint_state, alloc_soa, oracle_io

    def make(B):
        arr = alloc_soa(model, B)
        for k in ("q", "v", "command"):
            arr[k][:] = states[k][:, :B]
        if constraint_options is not None:
            alloc_constraint_state(model, arr, B)
        return arr

    def OracleEngineFor(arr):
        e = OracleEngine(model)
        if constraint_options is not None:
            e.set_constraint_options(**constraint_options)
            e.bind_constraints(arr["con_flags"], arr["con_data"])
        return e

    n_lanes, n_steps = 512, 4
    arr = make(n_lanes)
    e = OracleEngineFor(arr)
    io = oracle_io(arr)
    e.batch_run("start", io)
    e.batch_run("step", io, solver=solver, dt=dt, n_substeps=1, command_changed=False)
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < budget_s * 0.5:
        for _ in range(n_steps):
            e.batch_run("step", io, solver=solver, dt=dt, n_substeps=1,
                        command_changed=False)
        done += n_lanes * n_steps
    single = done / (time.perf_counter() - t0)

    # all cores: one PROCESS per core (no GIL, no shared allocator), each stepping its own slice
    cores = os.cpu_count() or 1
    per = 256
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    q_out = ctx.Queue()
    start_evt = ctx.Event()
    span = budget_s * 0.5

    def work(k):
        a = make(per)
        eng = OracleEngineFor(a)
        i = oracle_io(a)
        eng.batch_run("start", i)
        eng.batch_run("step", i, solver=solver, dt=dt, n_substeps=1, command_changed=False)
        start_evt.wait()
        n, t_begin = 0, time.perf_counter()
        while time.perf_counter() - t_begin < span:
            eng.batch_run("step", i, solver=solver, dt=dt, n_substeps=1, command_changed=False)
            n += per
        q_out.put((n, time.perf_counter() - t_begin))

    procs = [ctx.Process(target=work, args=(k,)) for k in range(cores)]
    for pr in procs:
        pr.start()
    time.sleep(1.0)            # let every worker finish its warm-up before the common start
    start_evt.set()
    res = [q_out.get() for _ in procs]
    for pr in procs:
        pr.join()
    multi = sum(n / t for n, t in res)
    return {
        "value": single, "unit": "env-steps/s", "cores": 1, "kind": "port",
        "sample": f"{n_lanes} lanes of the same seeded ANYmal batch stepped for ~{budget_s * 0.5:.0f} s "
                  f"by oracle/liboracle.so (g++ -O3, float64, {solver} dt={dt}"
                  + (", constraint contact model" if constraint_options is not None else "") + ")",
        "all_cores": {"value": multi, "cores": cores,
                      "sample": f"{cores} processes x {per} lanes, independent slices, "
                                f"~{budget_s * 0.5:.0f} s"},
    }


FULL_EXTRA_OUTPUTS = ("contact_forces", "energy", "joint_forces", "centroidal")


class _Ctx:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def measure(ctx, *, model_name: str, B: int, dtype, solver: str, contact_model: str, dt: float, steps: int, warmup: int,
            episode: int, extra_terms: str, gather_obs: bool = False, strong: bool = False, headline: bool = False):
    """One timed workload on this rank's GPU: `warmup` untimed steps, then exactly `steps` steps between barriers
    (wall clock, max over ranks -> `value`) with every step launch timed by HIP events on the launch stream inside the
    library (-> `roofline`).  Returns (JSON dict, the seeded states, the model)."""
    import torch
    import torch.distributed as dist

    from jiminy_amd import load_builtin
    from jiminy_amd.engine import BatchedEngine
    from jiminy_amd.synthetic import sample_standing_states, sample_states

    rank, world, device = ctx.rank, ctx.world, ctx.device
    model = load_builtin(model_name)
    constrained = contact_model == "constraint"
    dname = "f64" if dtype == torch.float64 else "f32"
    if constrained:
        # robots standing on all their feet (lowest contact point 5-6 mm into the ground, small joint /
        # attitude noise), 5 % of the lanes with joints beyond a position limit
        states = sample_standing_states(model, B, seed=rank, joint_noise=0.01, base_angle_max=0.004,
                                        depth_range=(-6e-3, -5e-3), twist_std=0.02, joint_vel_std=0.05,
                                        command_fraction=0.1, out_of_bounds_fraction=float(os.environ.get("JM_BENCH_OOB_FRACTION", "0.05")))
    else:
        states = sample_states(model, B, seed=rank)
    # extra_terms = "full": every optional output of the step is bound, so that the launch runs the whole of
    # Engine::computeExtraTerms (energies, subtree / centroidal quantities, the RNEA joint-wrench backward sweep)
    outputs = FULL_EXTRA_OUTPUTS if extra_terms == "full" else ("contact_forces",)
    eng = BatchedEngine(model, B, dtype=dtype, device=device, extra_outputs=outputs)
    eng.set_options({"stepper": {"odeSolver": solver, "dtMax": dt,
                                 "controllerUpdatePeriod": dt, "sensorsUpdatePeriod": dt},
                     "contacts": {"model": contact_model}})
    eng.set_command(torch.from_numpy(states["command"]).to(dtype))
    eng.start(torch.from_numpy(states["q"]).to(dtype), torch.from_numpy(states["v"]).to(dtype))
    # (untimed, outside the warm-up count: the first step of a simulation carries the reference's opening microsecond
    # step -- two launches, engine.substep_sizes --; every step after it is the periodic one the metric is quoted on)
    eng.step(dt)

    obs_rows = [eng.field(k) for k in ("imu", "force", "encoder", "effort") if eng._rows[k] > 0]
    gather = None
    if gather_obs and (world > 1 or getattr(ctx, "pg", False)):
        from jiminy_amd.distributed import ObservationGather
        gather = ObservationGather(dtype=torch.float32 if ctx.gather_dtype == "f32" else None, every=ctx.gather_every)
    gather_wait_s = 0.0
    q_seed = torch.from_numpy(states["q"]).to(dtype).to(device)
    v_seed = torch.from_numpy(states["v"]).to(dtype).to(device)
    all_lanes = torch.ones(B, dtype=torch.uint8, device=device)
    # lane status at every episode end: ONE device-to-device copy into a log inside the timed region; the statistics
    # (valid / NaN / out-of-bounds fractions) are computed from the log after it -- they are bookkeeping of the bench, not
    # part of the workload (round 5: the ~15 small reductions they cost per boundary were 5 % of a 20-step run)
    n_log = ((warmup + steps) // episode + 2) if episode > 0 else 1
    status_log = torch.zeros((n_log, B), dtype=torch.int32, device=device)
    n_logged = 0
    n_done = 0

    def one_step() -> None:
        nonlocal n_done, n_logged
        eng.step(dt)
        n_done += 1
        if episode > 0 and n_done % episode == 0:
            status_log[n_logged % n_log].copy_(eng.status.reshape(-1))
            n_logged += 1
            eng.reset_lanes(all_lanes, q_seed, v_seed)
        if gather is not None:
            # asynchronous: RCCL runs on the process group's stream behind an event of this stream; the
            # next step's launches overlap with it (the learner would call gather.result() where it reads)
            gather.launch(obs_rows)

    def barrier() -> None:
        if world > 1 or getattr(ctx, "pg", False):
            dist.barrier(device_ids=[ctx.local_rank])
        torch.cuda.synchronize(device)

    for _ in range(warmup):
        one_step()
    if episode > 0:
        # one untimed pass through the episode-boundary code (lazy torch kernels, reset launch)
        status_log[n_logged % n_log].copy_(eng.status.reshape(-1))
        n_logged += 1
        eng.reset_lanes(all_lanes, q_seed, v_seed)
        n_done = 0
    barrier()
    eng.enable_timing(os.environ.get("JM_BENCH_NO_EVENTS") != "1")   # (A/B of the cost of the per-launch HIP events: DESIGN.md section 12)
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    if gather is not None:
        # the timed region ends when the last gathered block has landed; how long that takes once the physics is done
        # is the part of the collectives the stepping did not hide (reported separately as `gather.exposed_ms`)
        torch.cuda.synchronize(device)
        tg = time.perf_counter()
        gather.drain()
        torch.cuda.synchronize(device)
        gather_wait_s = time.perf_counter() - tg
    barrier()
    elapsed = time.perf_counter() - t0
    n_launch, kernel_ms = eng.timing_summary()
    eng.enable_timing(False)
    if world > 1 or getattr(ctx, "pg", False):
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    status = eng.status.cpu().numpy()
    pgs_fail = float(((status & 16) != 0).mean())
    active = float((eng.field("con_flags") & 1).sum(0).double().mean().item()) if constrained else None
    status = status & ~16
    logged = (status_log[:min(n_logged, n_log)] & ~16).cpu().numpy()      # JM_LANE_SOLVER_FAILURE is not a lane failure
    ok_min = float((logged == 0).mean(axis=1).min()) if logged.size else 1.0
    nan_max = float(((logged & 1) != 0).mean(axis=1).max()) if logged.size else 0.0
    oob_max = float(((logged & 2) != 0).mean(axis=1).max()) if logged.size else 0.0
    ok_frac = min(float((status == 0).mean()), ok_min)
    nan_frac = max(float(((status & 1) != 0).mean()), nan_max)
    oob_frac = max(float(((status & 2) != 0).mean()), oob_max)
    # sanity of what the launch claims to compute: the full extra terms were written by the last step
    extras_written = None
    if extra_terms == "full":
        ok_l = torch.from_numpy(status == 0).to(device)
        e = eng.field("energy")[:, ok_l]
        jf = eng.field("joint_forces")[:, ok_l]
        extras_written = bool(ok_l.any().item() and torch.isfinite(e).all().item() and torch.isfinite(jf).all().item()
                              and float(e[0].abs().max().item()) > 0.0 and float(jf.abs().max().item()) > 0.0)
    del eng
    if rank != 0:
        return None, states, model

    value = world * B * steps / elapsed
    scal = algorithmic_scalars(model)
    sz = 8 if dname == "f64" else 4
    alg_bytes_per_launch = scal * sz * B
    avg_launch_s = (kernel_ms / max(n_launch, 1)) * 1e-3
    from jiminy_amd.codegen import quad_structure
    kernel_name = "jm::k_quad" if (quad_structure(model) is not None and
                                   os.environ.get("JM_KERNEL_VARIANT") != "lane") else "jm::k_batch"
    traffic = None
    valu = None
    # counters of the latest committed profile of this workload (tools/gpu_profile.sh -> profiles/)
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json" if model_name == "anymal" else f"pmc_{model_name}_latest.json")
    if constrained:
        # + the per-lane constraint state read and written once per step (flags int32, reference
        # configurations + multipliers); the delassus workspace is scratch, not algorithmic traffic
        rows = _abi_rows(model)
        alg_bytes_per_launch += 2 * (rows["con_flags"] * 4 + rows["con_data"] * sz) * B
        kernel_name = "jm::k_quad_con" if kernel_name == "jm::k_quad" else "jm::k_constrained"
        from jiminy_amd.codegen import qcon_split
        if kernel_name == "jm::k_quad_con" and qcon_split(model) and B % 16 == 0 and os.environ.get("JIMINY_AMD_QCON_SPLIT", "1") != "0":
            # large solves: one launch of the step = (k_quad_con_split<1> | solve | k_quad_con_split<2>) per evaluation; the solve
            # of robots with many contact points per foot runs in the operational space of the feet (k_qtip_pgs, jm_qtip.h),
            # k_qcon_pgs (the streamed form) only finds the robots that form could not take
            kernel_name = "jm::k_quad_con_split<1> + jm::k_qtip_pgs + jm::k_quad_con_split<2>"
            nbj = sum(1 for t in model.jtypes[1:] if 1 <= int(t) <= 8)
            if min(nbj + 4 * model.ncontacts, 96) <= 32:
                # robots with few contact points (round 6): the solve runs one lane per robot
                kernel_name = "jm::k_quad_con_split<1> + jm::k_qcon_pgs_lane + jm::k_quad_con_split<2>"
        pmc_path = os.path.join(ROOT, "profiles", "pmc_con_latest.json" if model_name == "anymal" else f"pmc_{model_name}_con_latest.json")
    achieved = alg_bytes_per_launch / avg_launch_s / 1e9 if n_launch else 0.0
    if os.path.exists(pmc_path):
        try:
            with open(pmc_path) as f:
                pmc = json.load(f)
            if pmc.get("batch") == B and pmc.get("model") == model_name and pmc.get("dtype") == dname \
                    and pmc.get("extra_terms", "sensors") == extra_terms:
                traffic = pmc.get("hbm_bytes_per_launch")
                # the roof that actually binds: VALU issue. Floor = every
                # VALU instruction of the profiled build issued back to back, waves spread
                # evenly over the 1024 SIMDs
                ipw, waves = pmc.get("valu_insts_per_wave"), pmc.get("waves_per_launch")
                if ipw and waves:
                    # cycles a SIMD needs per wave64 instruction, measured on the MI355X with two waves per SIMD
                    # (tools/microbench/op_issue, profiles/r05_op_issue.txt): 4.44 for the fp64 / 64-bit / DPP classes
                    # that make up the loop, 2.5 for plain 32-bit ops; clock: what the chip sustained under the profiled
                    # kernel (GRBM_GUI_ACTIVE / duration), nominal 2.4 GHz when the profile has none
                    cyc = 4.44 if dname == "f64" else 2.5
                    clock_hz = 1e9 * pmc["sustained_clock_ghz"] if pmc.get("sustained_clock_ghz") else CLOCK_HZ
                    floor_s = ipw * cyc * -(-int(waves) // SIMDS) / clock_hz
                    valu = {"bound": "valu-issue", "insts_per_wave_per_launch": ipw, "waves": waves,
                            "cycles_per_inst": cyc, "clock_ghz": clock_hz / 1e9, "floor_ms": 1e3 * floor_s,
                            "frac": floor_s / avg_launch_s if avg_launch_s else None,
                            # measured, not modelled: share of the time the VALU pipes of the profiled launch were issuing
                            "valu_pipe_busy_frac_profiled": pmc.get("valu_pipe_busy_frac"),
                            "from": pmc.get("from"),
                            "note": "counters come from the committed profile named in `from` (same workload, another run), "
                                    "only avg_launch_ms is measured in this run"}
        except Exception:
            traffic = None
    # fp64 vector roof (MI355X: 78.6 TFLOP/s = 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz; no MFMA on this path).  Two counts:
    # `executed` = measured VALU instructions per wave-launch (committed PMC profile) x flops per VALU instruction of the
    # evaluation loop (static ISA mix, tools/isa_mix.py) x 64 lanes x waves -- what the pipes did, replicated trunk work and
    # padding included; `algorithmic` = SURVEY.md 8d's estimate per env-step (47 kflop ANYmal, 114 kflop Atlas) x robots.
    flops = None
    mix_path = os.path.join(ROOT, "profiles", f"isa_mix_{model_name}_latest.json")
    alg_kflop = {"anymal": 47.0, "atlas": 114.0}.get(model_name)
    if valu is not None and os.path.exists(mix_path) and not constrained and solver == "runge_kutta_4":
        try:
            with open(mix_path) as f:
                mix = json.load(f)
            executed = valu["insts_per_wave_per_launch"] * mix["flops_per_valu_instruction_per_lane"] * 64.0 * valu["waves"]
            flops = {"bound": "fp64-valu", "peak": 78.6, "unit": "TFLOP/s",
                     "achieved_TFLOPs": executed / avg_launch_s / 1e12, "frac": executed / avg_launch_s / 1e12 / 78.6,
                     "flops_per_launch_executed": executed,
                     "achieved_TFLOPs_algorithmic": (alg_kflop * 1e3 * B / avg_launch_s / 1e12) if alg_kflop else None,
                     "frac_algorithmic": (alg_kflop * 1e3 * B / avg_launch_s / 1e12 / 78.6) if alg_kflop else None,
                     "isa_mix": {k: mix[k] for k in ("loop_valu", "loop_fma64", "loop_mul64", "loop_add64", "loop_dpp",
                                                     "fused_share_of_fp64_arith", "fp64_arith_share_of_valu")},
                     "from": [os.path.relpath(mix_path, ROOT), valu.get("from")]}
        except Exception:
            flops = None
    what_ran = ("4 dynamics evaluations (FK + contacts + motors + ABA)" if solver == "runge_kutta_4" else
                "1 dynamics evaluation (FK + contacts + motors + " + ("CRBA-free constrained ABA + PGS" if constrained else "ABA") + ")")
    extras_txt = ("full computeExtraTerms (energies, subtree masses / centroidal momentum and its derivative, RNEA joint "
                  "wrenches; + contact forces) + sensors" if extra_terms == "full"
                  else "sensor-level extra terms only (no energy / centroidal / joint-wrench sweep) + sensors")
    out = {
        "metric": ("env-steps/s (whole node) ANYmal 18-DoF batch 65536; achieved HBM GB/s vs peak"
                   if model_name == "anymal" and not constrained
                   else f"env-steps/s (whole node) {model_name} batch {B}"
                        + (" constraint contact model" if constrained else "")),
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "n_ranks_rccl": ctx.n_ranks,
        "dtype": dname, "data": "synthetic",
        "config": {"workload": f"{model_name} nq{model.nq} nv{model.nv} {model.nmotors} motors "
                               f"{model.ncontacts} {'constraint (PGS)' if constrained else 'spring-damper'} contact points, "
                               f"{solver} dt={dt} command held: {what_ran}, then {extras_txt}",
                   "extra_terms": extra_terms, "extra_terms_written": extras_written,
                   "lanes_per_gpu": B, "global_batch": world * B,
                   "parallelism": f"batch-sharded x{world}, no data-path collective"
                                  + (" + async obs all-gather (RCCL)" if gather is not None else ""),
                   "episode_steps": episode,
                   "lanes_ok_min": ok_frac, "lanes_nan_max": nan_frac,
                   "lanes_out_of_joint_bounds_max": oob_frac,
                   **({"mean_active_constraints": active, "lanes_pgs_iteration_cap_last_eval": pgs_fail}
                      if constrained else {})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_source": (f"PMC counters of the committed profile {os.path.relpath(pmc_path, ROOT)} (same workload, "
                                        "separate rocprofv3 --pmc passes; not re-measured by this run)") if traffic is not None else None,
                     "kernel": kernel_name, "launches_timed": n_launch,
                     "avg_launch_ms": 1e3 * avg_launch_s,
                     "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                     "secondary": valu, "flops": flops},
    }
    if gather is not None:
        out["gather"] = {"dtype": ctx.gather_dtype, "every": ctx.gather_every, "collectives": gather.launched,
                         "bytes_per_rank_per_collective": gather.bytes_per_rank,
                         "exposed_ms": 1e3 * gather_wait_s,
                         "note": "asynchronous all-gather on RCCL's stream, overlapped with the next steps; exposed_ms = "
                                 "wait for the collectives still in flight when the last step has finished"}
    if not headline:
        # compact form of a secondary workload
        out = {"workload": out["config"]["workload"], "model": model_name, "batch": B, "solver": solver, "dt": dt,
               "contact_model": contact_model, "extra_terms": extra_terms, "value": value, "unit": "env-steps/s",
               "steps": steps, "warmup": warmup, "ms_per_step": out["ms_per_step"],
               "ms_per_launch": 1e3 * avg_launch_s, "launches_timed": n_launch, "kernel": kernel_name,
               "algorithmic_bytes_per_launch": alg_bytes_per_launch, "achieved_GBps": achieved,
               "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "valu_issue": valu, "flops": flops,
               "lanes_nan_max": nan_frac,
               "lanes_ok_min": ok_frac, "extra_terms_written": extras_written,
               **({"mean_active_constraints": active, "lanes_pgs_iteration_cap_last_eval": pgs_fail} if constrained else {})}
    return out, states, model


# secondary workloads of the driver-run line (N = 1): the reference's SHIPPED configuration (anymal_options.toml:5,24 /
# atlas_options.toml: euler_explicit + contacts.model = "constraint") and BASELINE.json configs[3]'s robot
SECONDARY = (
    # the headline robot and solver at a QUARTER of the step: the headline's dt = 1e-3 with contacts.stiffness = 1e6 sits at the
    # edge of RK4's stability region (0.2 % of the lanes go non-finite within a 20-step episode, `lanes_nan_max` of the headline;
    # 0.03 % at 5e-4); at 2.5e-4 every lane stays finite and inside its bounds -- the same launch at the same cost, on a workload
    # the reference would not abort on
    dict(model_name="anymal", B=65536, solver="runge_kutta_4", contact_model="spring_damper", dt=2.5e-4, steps=20, warmup=3),
    dict(model_name="anymal", B=65536, solver="euler_explicit", contact_model="constraint", dt=1e-3, steps=20, warmup=3),
    dict(model_name="atlas", B=32768, solver="runge_kutta_4", contact_model="spring_damper", dt=2.5e-4, steps=20, warmup=3),
    dict(model_name="atlas", B=32768, solver="euler_explicit", contact_model="constraint", dt=5e-4, steps=8, warmup=2),
    # a seven-joint fixed-base arm (tests/data/arm7.urdf compiled into jiminy_amd/data/models/arm7.json): the class of robot
    # the one-robot-per-lane kernels serve
    dict(model_name="arm7", B=65536, solver="runge_kutta_4", contact_model="spring_damper", dt=1e-3, steps=20, warmup=3),
)


def measure_adaptive(ctx, *, model_name: str = "anymal", B: int = 65536, interval: float = 0.01, intervals: int = 4):
    """The reference's DEFAULT solver (`runge_kutta_dopri`, engine.h:307, one step size per robot) on the headline robot: robots
    landing on the spring-damper ground, default tolerances, `intervals` controller periods of `interval` seconds, one
    persistent launch each (jm_qdopri.h).  Unit: robot-intervals/s (a robot takes ~6 attempts of seven evaluations per
    interval on this workload); no roofline object -- the launch is bound by the sequential chain of its stiffest robot."""
    import torch

    from jiminy_amd import load_builtin
    from jiminy_amd.engine import BatchedEngine
    from jiminy_amd.synthetic import sample_states
    model = load_builtin(model_name)
    st = sample_states(model, B, seed=ctx.rank)
    eng = BatchedEngine(model, B, dtype=torch.float64, device=ctx.device)
    eng.set_options({"stepper": {"odeSolver": "runge_kutta_dopri", "controllerUpdatePeriod": interval, "sensorsUpdatePeriod": interval},
                     "contacts": {"model": "spring_damper"}})
    eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    eng.step(interval)          # (leaves the 1 us initial step size behind)
    torch.cuda.synchronize(ctx.device)
    it0 = eng.stepper_state.iter_lanes.double().mean().item()
    attempts, t0 = 0, time.perf_counter()
    for _ in range(intervals):
        eng.step(interval)
        attempts += eng.adaptive_attempts
    torch.cuda.synchronize(ctx.device)
    el = time.perf_counter() - t0
    ok = float(((eng.status.reshape(-1) & 9) == 0).double().mean().item())
    steps_acc = (eng.stepper_state.iter_lanes.double().mean().item() - it0) / intervals
    eng.stop()
    return {"workload": f"{model_name} runge_kutta_dopri spring_damper, {B} robots, {interval * 1e3:g} ms intervals, default tolerances",
            "model": model_name, "solver": "runge_kutta_dopri", "contact_model": "spring_damper", "batch": B,
            "metric": "robot-intervals/s", "value": B * intervals / el, "ms_per_interval": 1e3 * el / intervals,
            "attempts_of_the_stiffest_robot_per_interval": attempts / intervals, "mean_accepted_steps_per_interval": steps_acc,
            "lanes_ok": ok}


def secondary_workloads(ctx, args):
    import torch
    res = []
    for cfg in SECONDARY:
        try:
            out, _, _ = measure(ctx, dtype=torch.float64, episode=args.episode, extra_terms=args.extra_terms, **cfg)
            res.append(out)
        except Exception as e:  # a secondary workload must never take the headline down with it
            res.append({"workload": f"{cfg['model_name']} {cfg['contact_model']} {cfg['solver']}", "error": repr(e)[:300]})
        torch.cuda.empty_cache()
    try:
        res.append(measure_adaptive(ctx))
    except Exception as e:
        res.append({"workload": "anymal runge_kutta_dopri spring_damper", "error": repr(e)[:300]})
    torch.cuda.empty_cache()
    return res



def dry_run(args, rank: int, world: int) -> None:
    """The N > 1 control path without a GPU: rendezvous, barrier, the asynchronous observation gather and
    the max-over-ranks timing on CPU tensors over gloo; prints the same JSON shape with `value` null."""
    import torch
    import torch.distributed as dist

    from jiminy_amd import load_builtin
    from jiminy_amd.distributed import ObservationGather, shard_range
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n_ranks = dist.get_world_size() if world > 1 else 1
    if n_ranks != args.gpus:
        raise SystemExit(f"process group has {n_ranks} ranks, --gpus {args.gpus} requested")
    model = load_builtin(args.model)
    B = (shard_range(args.batch, rank, world)[1] - shard_range(args.batch, rank, world)[0]) if args.strong else args.batch
    B = min(B, 512)
    rows = [torch.full((6, B), float(rank), dtype=torch.float64), torch.full((2 * model.nmotors, B), float(rank), dtype=torch.float64)]
    gather = ObservationGather(dtype=torch.float32 if args.gather_dtype == "f32" else None,
                               every=args.gather_every) if (args.gather_obs and world > 1) else None
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if gather is not None:
            gather.launch(rows)
    ok = True
    if gather is not None:
        g = gather.result()
        ok = all(bool((g[r] == float(r)).all()) for r in range(world))
        gather.drain()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        print(json.dumps({"metric": "dry run (no physics)", "value": None, "unit": "env-steps/s", "n_gpus": world,
                          "n_ranks_rccl": n_ranks, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "gather_ok": ok, "scaling": "strong" if args.strong else "weak",
                          "lanes_per_gpu": B, "ms_per_step": 1e3 * elapsed / max(args.steps, 1)}))
    if world > 1:
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=65536, help="lanes per GPU")
    ap.add_argument("--model", default="anymal")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--solver", default=None, help="default: runge_kutta_4 (euler_explicit with the constraint model)")
    ap.add_argument("--contact-model", default="spring_damper", choices=["spring_damper", "constraint"],
                    help="'constraint': secondary workload, the contact model the reference's shipped ANYmal "
                         "options select (joint bounds + contact points as constraints, PGS), robots standing "
                         "on four feet")
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--gather-obs", action="store_true",
                    help="all-gather the observation block over RCCL every step (config 4 topology); asynchronous, "
                         "overlapped with the next step")
    ap.add_argument("--gather-dtype", default="f64", choices=["f64", "f32"],
                    help="type of the packed observation block of --gather-obs (f32 halves the bytes over xGMI)")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="--gather-obs: gather only every k-th step (a learner acting every k physics steps)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --batch is the GLOBAL batch, sharded over the ranks (config 4: "
                         "--model atlas --batch 32768 --strong --gather-obs)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / collective path only (gloo on CPU tensors, no physics): "
                         "what the world-size-2 CPU test of the N > 1 path runs")
    ap.add_argument("--episode", type=int, default=20,
                    help="steps between all-lane resets to the seeded states (0 = never). The reference "
                         "enforces joint bounds through its constraint solver (out of scope, DESIGN.md "
                         "section 8); random held torques drive lanes out of bounds after ~30 steps, so "
                         "the bench re-seeds like the vectorised env's auto-reset does, inside the timed "
                         "region, and reports the worst fraction of valid lanes seen before a reset")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra-terms", default="full", choices=["full", "sensors"],
                    help="'full' (default): energy, joint wrenches (the RNEA backward sweep), centroidal quantities, "
                         "fExternal and the contact forces are bound, so every launch runs the whole of "
                         "Engine::computeExtraTerms (engine.cc:800-905) like the reference does after every step; "
                         "'sensors': only what the sensors need (the round-3 headline)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workloads (ANYmal euler + constraint model, Atlas spring-damper, "
                         "Atlas constraint model) appended as `secondary` to the JSON line at N = 1")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(args.gpus))
    constrained = args.contact_model == "constraint"
    if args.solver is None:
        args.solver = "euler_explicit" if constrained else "runge_kutta_4"

    import torch
    import torch.distributed as dist

    from jiminy_amd import load_builtin
    from jiminy_amd.engine import BatchedEngine
    from jiminy_amd.synthetic import sample_standing_states, sample_states

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no GPU for LOCAL_RANK={local_rank} ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    n_ranks = 1
    # one process per GPU over RCCL.  A single rank started by torch.distributed.run with --gather-obs goes through the
    # same code -- communicator, barrier, max-over-ranks all-reduce, asynchronous all-gather -- with a world of one: the
    # only form of the N > 1 path a one-GPU box can execute (tests/test_multi_gpu.py)
    pg = world > 1 or ("WORLD_SIZE" in os.environ and args.gather_obs)
    if pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        n_ranks = dist.get_world_size()
        if n_ranks != args.gpus:
            raise SystemExit(f"process group has {n_ranks} ranks, --gpus {args.gpus} requested")

    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    if args.strong:
        from jiminy_amd.distributed import shard_range
        if args.batch % world:
            raise SystemExit("--strong needs a global batch that is a multiple of the number of GPUs")
        lo, hi = shard_range(args.batch, rank, world)
        B = hi - lo
    else:
        B = args.batch
    ctx = _Ctx(rank=rank, world=world, local_rank=local_rank, device=device, n_ranks=n_ranks, pg=pg,
               gather_dtype=args.gather_dtype, gather_every=args.gather_every)
    out, states, model = measure(ctx, model_name=args.model, B=B, dtype=dtype, solver=args.solver,
                                 contact_model=args.contact_model, dt=args.dt, steps=args.steps, warmup=args.warmup,
                                 episode=args.episode, extra_terms=args.extra_terms, gather_obs=args.gather_obs,
                                 strong=args.strong, headline=True)
    if rank == 0:
        if world == 1 and not args.no_secondary and args.model == "anymal" and not constrained and args.dtype == "f64":
            out["secondary"] = secondary_workloads(ctx, args)
        if world == 1 and not args.no_cpu_baseline and args.model == "anymal":
            out["cpu_baseline"] = cpu_baseline(model, states, args.dt, solver=args.solver,
                                               constraint_options={} if constrained else None)
        print(json.dumps(out))
    if pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
