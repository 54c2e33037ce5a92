"""REFERENCE-PINNED parity of the pipeline blocks.

tests/golden/ref_blocks.npz holds outputs of the reference's OWN functions -- `integrate_zoh`, `pd_controller`,
`pd_adapter` (gym_jiminy/common/blocks/proportional_derivative_controller.py:23-260), `mahony_filter`
(mahony_filter.py:28-101), `apply_safety_limits` (motor_safety_limit.py:20-77), `compute_tilt_from_quat`
(utils/math.py:1046-1060) -- executed from the reference tree by tools/make_ref_block_fixtures.py (numba.jit
stubbed to the identity).  Three implementations are checked against them, one application per comparison from
the fixture's inputs:

  * oracle/blocks_numpy.py (the scalar restatement the other tests use as their checker)      -- CPU
  * the tensor programs of jiminy_amd/blocks.py                                              -- CPU
  * the `jm_block_*` HIP kernels behind include/jiminy_hip.h, through `HipBlocks`            -- GPU

Tolerances: 1e-13 relative to the largest magnitude of the compared array (the reference compiles with
`fastmath=True`; sums may be re-associated by either side).
"""
import os

import numpy as np
import pytest
import torch

from jiminy_amd import blocks

FIX = os.path.join(os.path.dirname(__file__), "golden", "ref_blocks.npz")
TOL = 1e-13


@pytest.fixture(scope="module")
def ref():
    return dict(np.load(FIX))


def close(got, want, tol=TOL):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    return float(np.abs(got - want).max()) <= tol * max(float(np.abs(want).max()), 1.0)


# ------------------------------------------------------------------ the oracle's restatement
def test_oracle_blocks_reproduce_the_reference(ref):
    from oracle import blocks_numpy as orc
    T, _, M, B = ref["pd_cs_in"].shape
    for t in range(T):
        for b in range(B):
            state = np.ascontiguousarray(ref["pd_cs_in"][t, :, :, b])
            out = np.zeros(M)
            orc.pd_controller(ref["pd_enc"][t, :, :, b], state, ref["pd_lo"], ref["pd_hi"], ref["pd_kp"], ref["pd_kd"],
                              ref["pd_lim"], float(ref["pd_dt"][t]), out)
            assert close(state, ref["pd_cs_out"][t, :, :, b]) and close(out, ref["pd_out"][t, :, b])
    for k in range(len(ref["ad_order"])):
        for b in range(B):
            state = np.ascontiguousarray(ref["ad_cs_in"][k, :, :, b])
            out = np.zeros(M)
            orc.pd_adapter(ref["ad_action"][k, :, b].copy(), int(ref["ad_order"][k]), state, ref["pd_lo"], ref["pd_hi"],
                           bool(ref["ad_inst"][k]), ref["ad_db"] if ref["ad_use_db"][k] else None,
                           float(ref["ad_step_dt"]), out)
            assert close(state, ref["ad_cs_out"][k, :, :, b]) and close(out, ref["ad_out"][k, :, b])
    for t in range(ref["mh_imu"].shape[0]):
        for b in range(B):
            q, bias = np.ascontiguousarray(ref["mh_q_in"][t, :, :, b]), np.ascontiguousarray(ref["mh_bias_in"][t, :, :, b])
            om, cf = np.zeros((3, 1)), np.zeros((3, 1))
            orc.mahony_filter(q, om, cf, ref["mh_imu"][t, :3, b][:, None], ref["mh_imu"][t, 3:, b][:, None], bias,
                              float(ref["mh_kp"]), float(ref["mh_ki"]), float(ref["mh_dt"]))
            for got, key in ((q, "mh_q_out"), (bias, "mh_bias_out"), (om, "mh_omega"), (cf, "mh_cf")):
                assert close(got, ref[key][t, :, :, b])
    for b in range(B):
        assert close(np.stack(orc.compute_tilt_from_quat(ref["tilt_q"][:, :, b])), ref["tilt_v"][:, :, b])
    for t in range(ref["sl_cmd"].shape[0]):
        for b in range(B):
            out = np.zeros(M)
            orc.apply_safety_limits(ref["sl_cmd"][t, :, b], ref["sl_enc"][t, 0, :, b], ref["sl_enc"][t, 1, :, b],
                                    ref["sl_kp"], ref["sl_kd"], ref["sl_lo"], ref["sl_hi"], ref["sl_vlim"], ref["pd_lim"], out)
            assert close(out, ref["sl_out"][t, :, b])


# ------------------------------------------------------------------ the tensor programs (whole batch at once)
def test_tensor_program_blocks_reproduce_the_reference(ref):
    tt = torch.from_numpy
    T, _, M, B = ref["pd_cs_in"].shape
    lo, hi = tt(ref["pd_lo"]), tt(ref["pd_hi"])
    for t in range(T):
        cs, out = tt(ref["pd_cs_in"][t].copy()), torch.zeros((M, B), dtype=torch.float64)
        blocks.pd_controller(tt(ref["pd_enc"][t]), cs, lo, hi, tt(ref["pd_kp"]), tt(ref["pd_kd"]), tt(ref["pd_lim"]),
                             float(ref["pd_dt"][t]), out)
        assert close(cs.numpy(), ref["pd_cs_out"][t]) and close(out.numpy(), ref["pd_out"][t])
    for k in range(len(ref["ad_order"])):
        cs, out = tt(ref["ad_cs_in"][k].copy()), torch.zeros((M, B), dtype=torch.float64)
        blocks.pd_adapter(tt(ref["ad_action"][k].copy()), int(ref["ad_order"][k]), cs, lo, hi, bool(ref["ad_inst"][k]),
                          tt(ref["ad_db"]) if ref["ad_use_db"][k] else None, float(ref["ad_step_dt"]), out)
        assert close(cs.numpy(), ref["ad_cs_out"][k]) and close(out.numpy(), ref["ad_out"][k])
    for t in range(ref["mh_imu"].shape[0]):
        q, bias = tt(ref["mh_q_in"][t].copy()), tt(ref["mh_bias_in"][t].copy())
        om, cf = torch.zeros_like(bias), torch.zeros_like(bias)
        imu = tt(ref["mh_imu"][t]).view(1, 6, B).permute(1, 0, 2)
        blocks.mahony_filter(q, om, cf, imu[:3], imu[3:], bias, float(ref["mh_kp"]), float(ref["mh_ki"]), float(ref["mh_dt"]))
        for got, key in ((q, "mh_q_out"), (bias, "mh_bias_out"), (om, "mh_omega"), (cf, "mh_cf")):
            assert close(got.numpy(), ref[key][t])
    assert close(torch.stack(blocks.compute_tilt_from_quat(tt(ref["tilt_q"]))).numpy(), ref["tilt_v"])


# ------------------------------------------------------------------ the HIP kernels, through the C ABI
def _raw_encoder(enc_motor_order, enc_idx):
    """Engine encoder field `[n_enc][2][B]` whose sensor `enc_idx[m]` holds the (position, velocity) of motor m."""
    _, M, B = enc_motor_order.shape
    raw = np.zeros((M, 2, B))
    raw[enc_idx] = enc_motor_order.transpose(1, 0, 2)
    return torch.from_numpy(raw.reshape(2 * M, B))


@pytest.mark.gpu
def test_hip_blocks_reproduce_the_reference(gpu_device, ref):
    from jiminy_amd import load_builtin
    from jiminy_amd.engine import BatchedEngine
    model = load_builtin("anymal")
    T, _, M, B = ref["pd_cs_in"].shape
    assert M == model.nmotors
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    enc_idx = ref["enc_idx"]
    tt = torch.from_numpy
    hb = blocks.HipBlocks(eng, tt(enc_idx), tt(ref["pd_lo"]), tt(ref["pd_hi"]), tt(ref["pd_kp"]), tt(ref["pd_kd"]),
                          tt(ref["pd_lim"]))
    dev = lambda x: tt(np.ascontiguousarray(x)).to(gpu_device)  # noqa: E731
    for t in range(T):
        eng.field("encoder").copy_(_raw_encoder(ref["pd_enc"][t], enc_idx))
        cs, out = dev(ref["pd_cs_in"][t]), torch.zeros((M, B), dtype=torch.float64, device=gpu_device)
        hb.pd_controller(cs, float(ref["pd_dt"][t]), out)
        assert close(cs.cpu().numpy(), ref["pd_cs_out"][t]) and close(out.cpu().numpy(), ref["pd_out"][t])
    for k in range(len(ref["ad_order"])):
        cs, out = dev(ref["ad_cs_in"][k]), torch.full((M, B), 7.0, dtype=torch.float64, device=gpu_device)
        hb.pd_adapter(dev(ref["ad_action"][k]), int(ref["ad_order"][k]), cs, bool(ref["ad_inst"][k]),
                      tt(ref["ad_db"]) if ref["ad_use_db"][k] else None, float(ref["ad_step_dt"]), out)
        assert close(cs.cpu().numpy(), ref["ad_cs_out"][k]) and close(out.cpu().numpy(), ref["ad_out"][k])
    for t in range(ref["mh_imu"].shape[0]):
        eng.field("imu").copy_(tt(ref["mh_imu"][t]))
        q, bias = dev(ref["mh_q_in"][t]), dev(ref["mh_bias_in"][t])
        om, cf = torch.zeros_like(bias), torch.zeros_like(bias)
        hb.mahony_filter(q, om, cf, bias, float(ref["mh_kp"]), float(ref["mh_ki"]), float(ref["mh_dt"]))
        for got, key in ((q, "mh_q_out"), (bias, "mh_bias_out"), (om, "mh_omega"), (cf, "mh_cf")):
            assert close(got.cpu().numpy(), ref[key][t])
    for t in range(ref["sl_cmd"].shape[0]):
        eng.field("encoder").copy_(_raw_encoder(ref["sl_enc"][t], enc_idx))
        out = torch.zeros((M, B), dtype=torch.float64, device=gpu_device)
        hb.motor_safety_limit(dev(ref["sl_cmd"][t]), tt(ref["sl_kp"]), tt(ref["sl_kd"]), tt(ref["sl_lo"]), tt(ref["sl_hi"]),
                              tt(ref["sl_vlim"]), out)
        assert close(out.cpu().numpy(), ref["sl_out"][t])


def test_fixture_generator_is_committed_and_names_the_reference_functions():
    """The fixture is only as good as its provenance: the generating script travels with it."""
    path = os.path.join(os.path.dirname(__file__), "..", "tools", "make_ref_block_fixtures.py")
    src = open(path).read()
    for name in ("integrate_zoh", "pd_controller", "pd_adapter", "mahony_filter", "apply_safety_limits",
                 "compute_tilt_from_quat"):
        assert name in src
