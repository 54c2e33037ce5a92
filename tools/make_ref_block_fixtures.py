"""Reference-pinned fixtures of the gym_jiminy pipeline blocks: EXECUTES THE REFERENCE'S OWN PYTHON.

The block kernels of the reference are plain Python functions under `numba.jit`.  numba is not installed here
and the modules around them import the compiled `jiminy_py.core`, so the modules cannot be imported -- but the
FUNCTIONS can be run: this script parses the reference files, takes the function definitions named below
(decorators included), and executes them with `numba.jit` stubbed to the identity.  Nothing of the reference is
copied into the repository: only seeded inputs and the outputs the reference's code produced for them, written
to tests/golden/ref_blocks.npz.

Run in the build container (needs /root/reference):   python tools/make_ref_block_fixtures.py

Consumers (tests/test_reference_blocks.py): oracle/blocks_numpy.py, the tensor programs of
jiminy_amd/blocks.py (CPU) and the `jm_block_*` HIP kernels (GPU) are each compared with these outputs, one
application per comparison from identical inputs (the ZOH integrator truncates: chained independent
trajectories would amplify round-off).
"""
from __future__ import annotations

import ast
import os
import sys
import types

import numpy as np

REF = os.environ.get("JIMINY_REFERENCE", "/root/reference")
COMMON = os.path.join(REF, "python/gym_jiminy/common/gym_jiminy/common")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_blocks.npz")

SOURCES = {
    "utils/math.py": ("compute_tilt_from_quat",),
    "blocks/proportional_derivative_controller.py": ("integrate_zoh", "pd_controller", "pd_adapter"),
    "blocks/mahony_filter.py": ("mahony_filter",),
    "blocks/motor_safety_limit.py": ("apply_safety_limits",),
}


def load_reference_functions() -> dict:
    """Namespace holding the reference's functions, compiled from the reference's files where they lie."""
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)          # `@nb.jit(nopython=True, ...)` -> identity
    ns: dict = {"np": np, "nb": nb, "EARTH_SURFACE_GRAVITY": 9.81}
    import typing
    ns.update({k: getattr(typing, k) for k in ("Optional", "Tuple", "Union", "List")})
    for rel, names in SOURCES.items():
        path = os.path.join(COMMON, rel)
        with open(path) as f:
            tree = ast.parse(f.read(), filename=path)
        # module-level constants the functions read (EARTH_SURFACE_GRAVITY in mahony_filter.py) come from the file too
        for node in tree.body:
            if isinstance(node, ast.Assign) and all(isinstance(t, ast.Name) for t in node.targets) \
                    and isinstance(node.value, ast.Constant):
                exec(compile(ast.Module([node], []), path, "exec"), ns)
        found = set()
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                exec(compile(ast.Module([node], []), path, "exec"), ns)
                found.add(node.name)
        missing = set(names) - found
        if missing:
            raise RuntimeError(f"{rel}: functions {sorted(missing)} not found")
    return ns


def main() -> None:
    ref = load_reference_functions()
    rg = np.random.default_rng(20260927)
    M, B = 12, 96
    out: dict = {}

    # ---- pd_controller (+ integrate_zoh): T applications; the state is chained BY THE REFERENCE
    lo = np.stack([-1.0 - rg.random(M), -5.0 - rg.random(M), -50.0 - 50 * rg.random(M)])
    hi = np.stack([1.0 + rg.random(M), 5.0 + rg.random(M), 50.0 + 50 * rg.random(M)])
    kp, kd, lim = 100 + 1000 * rg.random(M), 0.01 + 0.1 * rg.random(M), 20 + 60 * rg.random(M)
    dts = np.array([5e-3, 5e-3, 1e-3, 5e-3, 0.0, 2e-2, 5e-3, 5e-3])
    T = len(dts)
    cs = np.stack([(rg.random((M, B)) - 0.5) * 2.6, (rg.random((M, B)) - 0.5) * 13, (rg.random((M, B)) - 0.5) * 250])
    cs[0] = np.clip(cs[0], lo[0][:, None], hi[0][:, None])
    enc = (rg.random((T, 2, M, B)) - 0.5) * 4.0
    cs_in, cs_out, tau = np.zeros((T, 3, M, B)), np.zeros((T, 3, M, B)), np.zeros((T, M, B))
    for t in range(T):
        cs_in[t] = cs
        for b in range(B):
            state = np.ascontiguousarray(cs[:, :, b])
            o = np.zeros(M)
            ref["pd_controller"](np.ascontiguousarray(enc[t, :, :, b]), state, lo, hi, kp, kd, lim, float(dts[t]), o)
            cs[:, :, b], tau[t, :, b] = state, o
        cs_out[t] = cs
        cs[2] = (rg.random((M, B)) - 0.5) * 250          # a new target acceleration, like a PD adapter upstream
    out.update(pd_lo=lo, pd_hi=hi, pd_kp=kp, pd_kd=kd, pd_lim=lim, pd_dt=dts, pd_enc=enc, pd_cs_in=cs_in,
               pd_cs_out=cs_out, pd_out=tau)

    # ---- pd_adapter: every (order, instantaneous, deadband) combination
    combos = [(o, i, d) for o in (0, 1) for i in (False, True) for d in (False, True)]
    K = len(combos)
    db = np.full(M, 0.3)
    step_dt = 0.04
    action = (rg.random((K, M, B)) - 0.5) * 3.0
    a_cs_in = np.stack([np.stack([(rg.random((M, B)) - 0.5) * 2.0, (rg.random((M, B)) - 0.5) * 8.0,
                                  (rg.random((M, B)) - 0.5) * 100]) for _ in range(K)])
    a_cs_out, a_out = a_cs_in.copy(), np.zeros((K, M, B))
    for k, (order, inst, use_db) in enumerate(combos):
        for b in range(B):
            state = np.ascontiguousarray(a_cs_in[k, :, :, b])
            o = np.zeros(M)
            ref["pd_adapter"](action[k, :, b].copy(), order, state, lo, hi, inst, db if use_db else None, step_dt, o)
            a_cs_out[k, :, :, b], a_out[k, :, b] = state, o
    out.update(ad_order=np.array([c[0] for c in combos]), ad_inst=np.array([c[1] for c in combos]),
               ad_use_db=np.array([c[2] for c in combos]), ad_db=db, ad_step_dt=step_dt, ad_action=action,
               ad_cs_in=a_cs_in, ad_cs_out=a_cs_out, ad_out=a_out)

    # ---- mahony_filter: one IMU per environment (ANYmal / Atlas), state chained by the reference; every 7th
    # environment is at rest with a zero bias estimate -> the early return
    TM = 6
    imu = (rg.random((TM, 6, B)) - 0.5) * np.array([1, 1, 1, 20, 20, 20.0])[None, :, None]
    imu[:, :, ::7] = 0.0
    quat = rg.random((4, 1, B)) - 0.5
    quat /= np.linalg.norm(quat, axis=0, keepdims=True)
    bias = (rg.random((3, 1, B)) - 0.5) * 0.1
    bias[:, :, ::7] = 0.0
    mh_kp, mh_ki, mh_dt = 1.0, 0.1, 5e-3
    q_in, b_in = np.zeros((TM, 4, 1, B)), np.zeros((TM, 3, 1, B))
    q_out, b_out, om_out, cf_out = (np.zeros((TM, 4, 1, B)), np.zeros((TM, 3, 1, B)), np.zeros((TM, 3, 1, B)),
                                    np.zeros((TM, 3, 1, B)))
    for t in range(TM):
        q_in[t], b_in[t] = quat, bias
        for b in range(B):
            q1, b1 = np.ascontiguousarray(quat[:, :, b]), np.ascontiguousarray(bias[:, :, b])
            om, cf = np.zeros((3, 1)), np.zeros((3, 1))
            ref["mahony_filter"](q1, om, cf, imu[t, :3, b][:, None].copy(), imu[t, 3:, b][:, None].copy(), b1,
                                 mh_kp, mh_ki, mh_dt)
            quat[:, :, b], bias[:, :, b], om_out[t, :, :, b], cf_out[t, :, :, b] = q1, b1, om, cf
        q_out[t], b_out[t] = quat, bias
    out.update(mh_imu=imu, mh_kp=mh_kp, mh_ki=mh_ki, mh_dt=mh_dt, mh_q_in=q_in, mh_bias_in=b_in, mh_q_out=q_out,
               mh_bias_out=b_out, mh_omega=om_out, mh_cf=cf_out)
    # the helper on its own, two orientations per call like a two-IMU robot
    tq = rg.random((4, 2, B)) - 0.5
    tq /= np.linalg.norm(tq, axis=0, keepdims=True)
    tilt = np.stack([np.stack(ref["compute_tilt_from_quat"](np.ascontiguousarray(tq[:, :, b]))) for b in range(B)], -1)
    out.update(tilt_q=tq, tilt_v=tilt)

    # ---- apply_safety_limits
    TS = 3
    s_cmd = (rg.random((TS, M, B)) - 0.5) * 200
    s_enc = np.stack([(rg.random((TS, M, B)) - 0.5) * 2.4, (rg.random((TS, M, B)) - 0.5) * 20], 1)   # [TS][2][M][B]
    s_kp, s_kd = 20 + 80 * rg.random(M), 0.5 + 2 * rg.random(M)
    s_lo, s_hi = -1.0 + 0.2 * rg.random(M), 1.0 - 0.2 * rg.random(M)
    s_vlim = 5 + 3 * rg.random(M)
    s_out = np.zeros((TS, M, B))
    for t in range(TS):
        for b in range(B):
            o = np.zeros(M)
            ref["apply_safety_limits"](s_cmd[t, :, b].copy(), s_enc[t, 0, :, b].copy(), s_enc[t, 1, :, b].copy(), s_kp,
                                       s_kd, s_lo, s_hi, s_vlim, lim, o)
            s_out[t, :, b] = o
    out.update(sl_cmd=s_cmd, sl_enc=s_enc, sl_kp=s_kp, sl_kd=s_kd, sl_lo=s_lo, sl_hi=s_hi, sl_vlim=s_vlim, sl_out=s_out)

    out["enc_idx"] = rg.permutation(M)          # sensor -> motor permutation the device tests bind the encoder field with
    np.savez_compressed(OUT, **out)
    n_sat = int((np.abs(tau) == lim[None, :, None]).sum())
    print(f"wrote {os.path.relpath(OUT)}: {sum(v.nbytes for v in out.values() if hasattr(v, 'nbytes')) / 1e6:.2f} MB raw, "
          f"{n_sat} saturated PD outputs, {int((s_out != s_cmd).sum())} clipped safety outputs")


if __name__ == "__main__":
    if not os.path.isdir(COMMON):
        sys.exit(f"{COMMON} not found: run this where the reference tree is available")
    main()
