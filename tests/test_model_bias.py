"""Model biases per environment drawn from the reference's generator stream (model part):
`Model::addBiasedToExtendedModel` (core/src/robot/model.cc:1166-1236) from `Engine::generator_` seeded with
`std::seed_seq{seed}` (core/src/engine/engine.cc:756-757, utilities/random.hxx:20-51).

CPU: the oracle restatement (oracle/oracle_random.cpp `orc_model_bias`, `orc_engine_rng_seed`) is pinned on
independent restatements -- the seeding on the [rand.util.seedseq] algorithm in Python integers, the draws on the
oracle's own normal stream replayed in the reference's joint / field order with the arithmetic redone in numpy.
GPU: `jm_block_model_bias` / `jm_engine_rng_seed` through the engine against the oracle, generator states bit for bit.
"""
import numpy as np
import pytest

from jiminy_amd import load_builtin
from jiminy_amd.randomization import nominal_bias_table, nominal_model_lane
from oracle import oracle_py

from tests.test_sensor_noise import py_seed_seq

STD = {"inertiaBodiesBiasStd": 0.05, "massBodiesBiasStd": 0.1, "centerOfMassPositionBodiesBiasStd": 0.04,
       "relativePositionBodiesBiasStd": 0.02}
ORDER = ("inertiaBodiesBiasStd", "massBodiesBiasStd", "centerOfMassPositionBodiesBiasStd", "relativePositionBodiesBiasStd")


def _std4(d):
    return np.array([d.get(k, 0.0) for k in ORDER], dtype=np.float32)


@pytest.mark.parametrize("seed", [0, 1, 123456789, 0xFFFFFFFF])
def test_engine_generator_seeding_is_generate_state_of_a_seed_seq(seed):
    """internal::generateState (random.hxx:20-44): two words of std::seed_seq{seed}, low word first, | 3."""
    w = py_seed_seq([seed], 2)
    want = ((int(w[0]) | (int(w[1]) << 32)) | 3) & 0xFFFFFFFFFFFFFFFF
    got = oracle_py.engine_rng_seed(np.array([seed], dtype=np.uint32))
    assert int(got[0]) == want


def _exp3(v):
    t = np.linalg.norm(v)
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    if t < 1e-8:
        return np.eye(3) + K
    return np.eye(3) + np.sin(t) / t * K + (1 - np.cos(t)) / t ** 2 * (K @ K)


@pytest.mark.parametrize("name", ["anymal", "double_pendulum"])
def test_oracle_model_bias_consumes_the_stream_in_the_reference_order(name):
    """Per mechanical joint (the free-flyer root is none, model.cc:337-341): com (3 normals), mass (1), inertia (3 for the
    rotation vector + 3 for the moments), relative position (3), each group only when enabled."""
    model = load_builtin(name)
    nj, first = model.njoints, (2 if model.has_freeflyer else 1)
    nom = nominal_bias_table(model)
    B = 5
    for opts in (STD, {"massBodiesBiasStd": 0.3}, {"inertiaBodiesBiasStd": 0.2, "relativePositionBodiesBiasStd": 0.1}):
        s4 = _std4(opts)
        rng = oracle_py.engine_rng_seed(np.arange(B, dtype=np.uint32) + 11)
        rng0 = rng.copy()
        out = nominal_model_lane(model, B).numpy().copy()
        oracle_py.model_bias(nom, first, s4, rng, out)
        per_joint = 3 * (s4[2] > 0) + 1 * (s4[1] > 0) + 6 * (s4[0] > 0) + 3 * (s4[3] > 0)
        for l in range(B):
            normals, state = oracle_py.pcg32_stream(int(rng0[l]), int(per_joint) * (nj - first), "normal")
            # the generator ends exactly where that many normals leave it ...
            assert state == int(rng[l])
            it = iter(normals)
            nxt = lambda mean, std: np.float64(np.float32(np.float32(next(it)) * np.float32(std)) + np.float32(mean))  # noqa: E731
            for j in range(first, nj):
                got = out[13 * j:13 * j + 13, l]
                com, mass, pos = nom[j, 1:4].copy(), nom[j, 0], nom[j, 10:13].copy()
                I = np.array([[nom[j, 4], nom[j, 5], nom[j, 6]], [nom[j, 5], nom[j, 7], nom[j, 8]], [nom[j, 6], nom[j, 8], nom[j, 9]]])
                if s4[2] > 0:
                    com = com * np.array([nxt(1.0, s4[2]) for _ in range(3)])
                if s4[1] > 0:
                    mass = max(mass * nxt(1.0, s4[1]), min(mass, 1e-3))
                if s4[0] > 0:
                    ra = np.array([nxt(0.0, s4[0]) for _ in range(3)])
                    A = nom[j, 16:25].reshape(3, 3) @ _exp3(ra)
                    M = nom[j, 13:16] * np.array([nxt(1.0, s4[0]) for _ in range(3)])
                    I = A @ np.diag(M) @ A.T
                if s4[3] > 0:
                    pos = pos * np.array([nxt(1.0, s4[3]) for _ in range(3)])
                # ... and the values are the reference's arithmetic on those normals
                assert got[0] == mass
                np.testing.assert_array_equal(got[1:4], com)
                np.testing.assert_array_equal(got[10:13], pos)
                np.testing.assert_allclose(got[4:10], [I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]], rtol=1e-12, atol=1e-18)
        # the root body (and the universe) keep the nominal parameters
        np.testing.assert_array_equal(out[:13 * first], nominal_model_lane(model, B).numpy()[:13 * first])


def test_oracle_model_bias_laws_and_masked_redraw():
    model = load_builtin("anymal")
    B = 8192
    nom = nominal_bias_table(model)
    rng = oracle_py.engine_rng_seed(np.arange(B, dtype=np.uint32))
    out = nominal_model_lane(model, B).numpy().copy()
    oracle_py.model_bias(nom, 2, _std4(STD), rng, out)
    ml = out.reshape(model.njoints, 13, B)
    j = 3
    r = ml[j, 0] / nom[j, 0]
    assert r.mean() == pytest.approx(1.0, abs=5e-3) and r.std() == pytest.approx(0.1, rel=5e-2)
    assert (ml[j, 1:4] / nom[j, 1:4, None]).std() == pytest.approx(0.04, rel=5e-2)
    I = ml[j, 4:10]
    M = np.stack([np.stack([I[0], I[1], I[2]]), np.stack([I[1], I[3], I[4]]), np.stack([I[2], I[4], I[5]])]).transpose(2, 0, 1)
    ev = np.linalg.eigvalsh(M)
    assert ev.min() > 0.0 and np.allclose(ev.mean(0), nom[j, 13:16], rtol=5e-2)
    # episode-wise re-draw: only the masked lanes change, the generators of the others stay where they were
    mask = np.zeros(B, dtype=np.uint8)
    mask[::3] = 1
    rng1, out1 = rng.copy(), out.copy()
    oracle_py.model_bias(nom, 2, _std4(STD), rng1, out1, mask)
    keep = mask == 0
    assert np.array_equal(out1[:, keep], out[:, keep]) and np.array_equal(rng1[keep], rng[keep])
    assert not np.array_equal(out1[:, ~keep], out[:, ~keep]) and np.all(rng1[~keep] != rng[~keep])
    # a tiny mass keeps its floor: max(m * N(1, std), min(m, 1 g))
    light = nom.copy()
    light[4, 0] = 5.0e-4
    rng = oracle_py.engine_rng_seed(np.arange(4096, dtype=np.uint32))
    out = np.zeros((13 * model.njoints, 4096))
    oracle_py.model_bias(light, 2, _std4({"massBodiesBiasStd": 0.5}), rng, out)
    assert out[13 * 4].min() >= 5.0e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype_name", [("anymal", "float64"), ("atlas", "float64"), ("anymal", "float32")])
def test_engine_model_biases_follow_the_reference_stream(gpu_device, name, dtype_name):
    """`BatchedEngine.sample_model_biases` (`jm_block_model_bias`) against the oracle: generator states bit for bit (every
    accept / reject decision of the ziggurat agreed), values equal where the float normal is (fast path: ~99 % of the
    draws) and within one float ulp of the normal elsewhere (device logf / expf in the wedge and tail samples)."""
    import torch

    from jiminy_amd.engine import BatchedEngine
    model = load_builtin(name)
    dtype = getattr(torch, dtype_name)
    B = 4099
    eng = BatchedEngine(model, B, dtype=dtype, device=gpu_device)
    eng.set_model_options({"dynamics": STD})
    seeds = (np.arange(B, dtype=np.uint64) * 7919 + 5).astype(np.uint32)
    eng.seed_model(seeds)
    rng = oracle_py.engine_rng_seed(seeds)
    assert np.array_equal(eng.model_rng_state.cpu().numpy().view(np.uint64), rng)
    nom = nominal_bias_table(model)
    first = 2 if model.has_freeflyer else 1
    ref = nominal_model_lane(model, B).numpy().copy()
    mask = None
    for rnd in range(3):     # full draw, then two masked re-draws (lanes being reset)
        if rnd:
            mask = (np.arange(B) % (rnd + 2) == 0)
        ml = eng.sample_model_biases(None if mask is None else torch.from_numpy(mask).to(gpu_device)).cpu().numpy().astype(np.float64)
        oracle_py.model_bias(nom, first, _std4(STD), rng, ref, None if mask is None else mask.astype(np.uint8))
        assert np.array_equal(eng.model_rng_state.cpu().numpy().view(np.uint64), rng), rnd
        want = ref if dtype_name == "float64" else ref.astype(np.float32).astype(np.float64)
        # entries of one body's inertia are compared on the scale of that inertia: a tail / wedge normal that differs by
        # one float ulp in the rotation vector moves the (cancellation-dominated) off-diagonal entries by 1e-11 absolute
        scale = np.abs(want).copy()
        w3 = scale.reshape(model.njoints, 13, B)
        w3[:, 4:10] = w3[:, 4:10].max(axis=1, keepdims=True)
        rel = np.abs(ml - want) / np.maximum(scale, 1e-300)
        rel[scale == 0.0] = np.abs(ml[scale == 0.0])
        tol = 4e-7 if dtype_name == "float64" else 6e-7
        assert rel.max() < tol, (rnd, rel.max())
        rows = np.r_[[13 * j + i for j in range(first, model.njoints) for i in (0, 1, 2, 3, 10, 11, 12)]]
        exact = (ml[rows] == want[rows]).mean()
        assert exact > 0.97, exact
