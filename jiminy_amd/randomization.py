"""Per-environment model biases on the device: one randomised copy of the robot's body parameters per
lane, laid out as the `JM_F_MODEL_LANE` field (`[13 * njoints][B]`: mass | com 3 | inertia xx xy xz yy yz zz |
joint placement translation 3 per joint).

Restates `Model::addBiasedToExtendedModel` (reference core/src/robot/model.cc:1166-1236) for B robots at
once, with the model options of `Model::getDefaultDynamicsOptions` (core/include/jiminy/core/robot/model.h:147-158):

* `centerOfMassPositionBodiesBiasStd`: every component of the body's centre of mass times N(1, std);
* `massBodiesBiasStd`: mass = max(mass * N(1, std), min(mass, 1 g));
* `inertiaBodiesBiasStd`: principal moments times N(1, std) each, principal axes rotated by exp3(N(0, std)^3)
  (eigen-decomposition of the rotational inertia, so that it stays positive semi-definite);
* `relativePositionBodiesBiasStd`: every component of the joint placement translation times N(1, std)
  (rotation excluded).

The engine draws on the device with `jm_block_model_bias` (csrc/jm_random.h): one PCG32 stream per lane, consumed in
the reference's joint and field order, generator states bit-identical to `oracle/oracle_random.cpp`
(`BatchedEngine.sample_model_biases`).  `sample_model_lane` below is the tensor-program statement of the same laws
from a `torch.Generator`: host-side use and the statistical tests of tests/test_variation.py.  The free-flyer root is
not a "mechanical joint" (model.cc:337-341): its body is never biased.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .model import CompiledModel

MODEL_LANE_ROWS = 13
DYNAMICS_OPTION_NAMES = ("inertiaBodiesBiasStd", "massBodiesBiasStd", "centerOfMassPositionBodiesBiasStd",
                         "relativePositionBodiesBiasStd")


def default_dynamics_options() -> Dict[str, float]:
    """≙ `Model::getDefaultDynamicsOptions` (model.h:147-158), the bias options."""
    return {k: 0.0 for k in DYNAMICS_OPTION_NAMES}


def nominal_model_lane(model: CompiledModel, batch_size: int, dtype: torch.dtype = torch.float64,
                       device: Optional[torch.device] = None) -> torch.Tensor:
    """The model's own body parameters replicated for every lane (row layout of JM_F_MODEL_LANE)."""
    nj = model.njoints
    rows = np.zeros((MODEL_LANE_ROWS * nj,))
    for j in range(1, nj):
        I = model.inertia[j]
        rows[13 * j] = model.mass[j]
        rows[13 * j + 1:13 * j + 4] = model.com[j]
        rows[13 * j + 4:13 * j + 10] = (I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2])
        rows[13 * j + 10:13 * j + 13] = model.placement_p[j]
    t = torch.as_tensor(rows, dtype=dtype, device=device)
    return t[:, None].expand(-1, batch_size).contiguous()


def nominal_bias_table(model: CompiledModel) -> np.ndarray:
    """`[njoints][25]` float64: mass | com 3 | inertia xx xy xz yy yz zz | placement translation 3 | principal moments 3
    (ascending) | principal axes 9 (row-major, columns = axes) of the unbiased bodies: the `nominal` argument of
    `jm_block_model_bias`.  (The reference decomposes with Eigen::SelfAdjointEigenSolver, model.cc:1204-1206; the sign
    and, for equal moments, the choice of the axes are the solver's, not part of the law.)"""
    nj = model.njoints
    t = np.zeros((nj, 25))
    for j in range(1, nj):
        I = np.asarray(model.inertia[j], dtype=np.float64)
        t[j, 0] = model.mass[j]
        t[j, 1:4] = model.com[j]
        t[j, 4:10] = (I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2])
        t[j, 10:13] = model.placement_p[j]
        w, A = np.linalg.eigh(0.5 * (I + I.T))
        t[j, 13:16] = w
        t[j, 16:25] = A.reshape(9)
    return np.ascontiguousarray(t)


def _exp3(w: torch.Tensor) -> torch.Tensor:
    """Rodrigues' formula for a batch of rotation vectors `[..., 3]` -> `[..., 3, 3]`."""
    t2 = (w * w).sum(-1)
    t = torch.sqrt(t2)
    small = t < 1e-8
    ts = torch.where(small, torch.ones_like(t), t)
    a = torch.where(small, 1.0 - t2 / 6.0, torch.sin(ts) / ts)
    b = torch.where(small, 0.5 - t2 / 24.0, (1.0 - torch.cos(ts)) / (ts * ts))
    K = torch.zeros(w.shape[:-1] + (3, 3), dtype=w.dtype, device=w.device)
    K[..., 0, 1], K[..., 0, 2] = -w[..., 2], w[..., 1]
    K[..., 1, 0], K[..., 1, 2] = w[..., 2], -w[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -w[..., 1], w[..., 0]
    eye = torch.eye(3, dtype=w.dtype, device=w.device).expand_as(K)
    return eye + a[..., None, None] * K + b[..., None, None] * (K @ K)


def sample_model_lane(model: CompiledModel, batch_size: int, options: Dict[str, float],
                      generator: Optional[torch.Generator] = None, dtype: torch.dtype = torch.float64,
                      device: Optional[torch.device] = None, lane_mask: Optional[torch.Tensor] = None,
                      previous: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Biased body parameters for every lane (`[13 * njoints][B]`).  `lane_mask` / `previous`: only the masked
    lanes get new draws (episode-wise re-randomisation of the environments being reset)."""
    dev = torch.device("cpu") if device is None else torch.device(device)
    nj, B = model.njoints, int(batch_size)
    f32 = torch.float32   # the reference draws float normals (`normal(g, 1.0F, std)`)

    def normal(shape, mean, std):
        x = torch.randn(shape, generator=generator, dtype=f32, device=generator.device if generator is not None else "cpu")
        return (mean + std * x).to(torch.float64).to(dev)
    out = nominal_model_lane(model, B, torch.float64, dev).view(nj, MODEL_LANE_ROWS, B).clone()
    com_std = float(options.get("centerOfMassPositionBodiesBiasStd", 0.0))
    mass_std = float(options.get("massBodiesBiasStd", 0.0))
    inertia_std = float(options.get("inertiaBodiesBiasStd", 0.0))
    pos_std = float(options.get("relativePositionBodiesBiasStd", 0.0))
    eps = float(np.finfo(np.float64).eps)
    for j in range(2 if model.has_freeflyer else 1, nj):   # mechanical joints: the free-flyer root is not one
        if com_std > eps:
            out[j, 1:4] *= normal((3, B), 1.0, com_std)
        if mass_std > eps:
            m0 = out[j, 0]
            out[j, 0] = torch.maximum(m0 * normal((B,), 1.0, mass_std), torch.minimum(m0, torch.full_like(m0, 1.0e-3)))
        if inertia_std > eps:
            I = torch.as_tensor(model.inertia[j], dtype=torch.float64, device=dev)
            moments, axes = torch.linalg.eigh(I)
            rot = _exp3(normal((B, 3), 0.0, inertia_std))                        # (B, 3, 3)
            axes_b = axes[None] @ rot                                             # A * R
            mom_b = moments[None] * normal((B, 3), 1.0, inertia_std)              # (B, 3)
            Ib = (axes_b * mom_b[:, None, :]) @ axes_b.transpose(1, 2)           # A diag(M) A^T
            out[j, 4], out[j, 5], out[j, 6] = Ib[:, 0, 0], Ib[:, 0, 1], Ib[:, 0, 2]
            out[j, 7], out[j, 8], out[j, 9] = Ib[:, 1, 1], Ib[:, 1, 2], Ib[:, 2, 2]
        if pos_std > eps:
            out[j, 10:13] *= normal((3, B), 1.0, pos_std)
    out = out.view(nj * MODEL_LANE_ROWS, B)
    if lane_mask is not None and previous is not None:
        out = torch.where(lane_mask.to(dev)[None, :], out, previous.to(torch.float64))
    return out.to(dtype).contiguous()
