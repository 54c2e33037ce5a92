"""Batched periodic Gaussian processes (jiminy_amd/processes.py) against the scalar restatement of the reference's
(oracle/process_numpy.py), and the properties the reference's process has by construction."""
import numpy as np
import pytest
import torch

from jiminy_amd.processes import PeriodicGaussianProcess, toeplitz_cholesky_lower
from oracle import process_numpy


@pytest.mark.parametrize("wavelength,period", [(0.2, 1.0), (1.0, 1.0), (0.35, 2.0)])
def test_batched_process_matches_the_scalar_restatement(wavelength, period):
    B = 5
    ref = process_numpy.PeriodicGaussianProcess(wavelength, period)
    proc = PeriodicGaussianProcess(wavelength, period, B)
    assert proc.num_times == ref.num_times and abs(proc.dt - ref.dt) < 1e-15
    assert np.allclose(proc._L.numpy(), np.tril(ref.cov_sqrt_root), rtol=0, atol=1e-7)   # (the recursion is ill-conditioned: reg = 1e-9)
    rng = np.random.default_rng(3)
    z = rng.standard_normal((ref.num_times, B)).astype(np.float32)
    proc.reset(normal=torch.from_numpy(z))
    times = np.concatenate([np.linspace(-0.7, 2.9 * period, 41), [0.0, period, ref.dt, period - 1e-12]])
    for lane in range(B):
        ref.reset(z[:, lane].astype(np.float64))
        assert np.allclose(proc.values[:, lane].numpy(), ref.values, rtol=0, atol=1e-6)
        assert np.allclose(proc.grads[:, lane].numpy(), ref.grads, rtol=1e-6, atol=1e-5)   # L is ill-conditioned by design (reg 1e-9)
        for t in times:
            assert abs(float(proc(float(t))[lane]) - ref(float(t))) < 1e-6
            assert abs(float(proc.grad(float(t))[lane]) - ref.grad(float(t))) < 1e-4
    # one time per lane
    tl = torch.tensor([0.05, 0.3, 0.99, 1.7, -0.2], dtype=torch.float64)
    got = proc(tl)
    for lane in range(B):
        ref.reset(z[:, lane].astype(np.float64))
        assert abs(float(got[lane]) - ref(float(tl[lane]))) < 1e-6


def test_process_is_periodic_smooth_and_standard_normal():
    B = 4096
    proc = PeriodicGaussianProcess(0.2, 1.0, B)
    g = torch.Generator().manual_seed(7)
    proc.reset(g)
    assert torch.allclose(proc(0.013), proc(1.013), atol=1e-9) and torch.allclose(proc(0.5), proc(-0.5), atol=1e-9)
    # unit variance at every knot (the covariance has a unit diagonal), zero mean
    assert abs(float(proc.values.mean())) < 0.02 and abs(float(proc.values.var()) - 1.0) < 0.05
    # covariance of neighbouring knots = the periodic squared-exponential kernel; knots half a period apart: none
    v = proc.values
    c1 = float((v[0] * v[1]).mean())
    c25 = float((v[0] * v[25]).mean())
    assert abs(c1 - float(np.exp(-2.0 * (np.sin(np.pi / 50) / 0.2) ** 2))) < 0.04 and abs(c25) < 0.1   # the kernel itself
    # the tabulated derivative is the derivative of the interpolant
    h = 1e-6
    t = 0.3141
    fd = (proc(t + h) - proc(t - h)) / (2 * h)
    assert torch.allclose(fd, proc.grad(t), atol=1e-4)
    # re-drawing masked lanes only
    before = proc.values.clone()
    mask = torch.zeros(B, dtype=torch.bool)
    mask[:10] = True
    proc.reset(g, lane_mask=mask)
    assert torch.equal(proc.values[:, 10:], before[:, 10:]) and not torch.equal(proc.values[:, :10], before[:, :10])


def test_toeplitz_factor_reproduces_the_covariance():
    n, wl = 50, 0.2
    i = np.arange(n)
    c = np.exp(-2.0 * (np.sin(np.pi / n * i) / wl) ** 2)
    L = toeplitz_cholesky_lower(c, 1e-9)
    K = np.array([[c[abs(a - b)] for b in range(n)] for a in range(n)])
    assert np.allclose(L @ L.T, K + 1e-9 * np.eye(n), atol=1e-7)
    with pytest.raises(ValueError):
        PeriodicGaussianProcess(0.2, -1.0, 2)


@pytest.mark.parametrize("wavelength,period", [(0.2, 1.0), (0.35, 2.0)])
def test_fourier_process_matches_the_scalar_restatement_and_its_series(wavelength, period):
    """`PeriodicFourierProcess` (random.cc:462-485): batched against scalar, and at the knots the tabulated values /
    derivatives ARE the truncated Fourier series and its time derivative."""
    import math

    from jiminy_amd.processes import PeriodicFourierProcess
    B = 4
    ref = process_numpy.PeriodicFourierProcess(wavelength, period)
    proc = PeriodicFourierProcess(wavelength, period, B)
    H, n = ref.num_harmonics, ref.num_times
    assert proc.num_harmonics == H == math.ceil(period / wavelength) and proc.num_times == n
    rng = np.random.default_rng(8)
    z = rng.standard_normal((2 * H, B)).astype(np.float32)
    proc.reset(normal=torch.from_numpy(z))
    scale = math.sqrt(2.0 / (2 * H + 1))
    for lane in range(B):
        z1, z2 = z[:H, lane].astype(np.float64), z[H:, lane].astype(np.float64)
        ref.reset(z1, z2)
        assert np.allclose(proc.values[:, lane].numpy(), ref.values, rtol=0, atol=1e-12)
        assert np.allclose(proc.grads[:, lane].numpy(), ref.grads, rtol=0, atol=1e-10)
        tk = np.arange(n) * ref.dt
        k = np.arange(1, H + 1)[None, :]
        series = scale * (np.sin(2 * np.pi * k * tk[:, None] / period) @ z1 + np.cos(2 * np.pi * k * tk[:, None] / period) @ z2)
        dseries = scale * ((np.cos(2 * np.pi * k * tk[:, None] / period) * (2 * np.pi * k / period)) @ z1
                           - (np.sin(2 * np.pi * k * tk[:, None] / period) * (2 * np.pi * k / period)) @ z2)
        assert np.allclose(ref.values, series, atol=1e-12) and np.allclose(ref.grads, dseries, atol=1e-10)
        for t in (-0.3, 0.0, 0.123, period, 1.7 * period):
            assert abs(float(proc(float(t))[lane]) - ref(float(t))) < 1e-10
            assert abs(float(proc.grad(float(t))[lane]) - ref.grad(float(t))) < 1e-8
    # unit variance by construction: 2 H amplitudes of variance scale^2 / 2 each ... = 2 H / (2 H + 1)
    big = PeriodicFourierProcess(wavelength, period, 8192)
    big.reset(torch.Generator().manual_seed(1))
    assert abs(float(big.values.var()) - 2 * H / (2 * H + 1)) < 0.05


def test_perlin_time_processes_match_the_scalar_restatement_lane_by_lane():
    """`RandomPerlinProcess<1>` / `PeriodicPerlinProcess<1>` (random.h:564-590) as per-lane time processes: every lane equals
    the scalar restatement reset from `PCG32(seed of the lane)` BIT FOR BIT (hash / table gradients, float32 draws, octave
    wavelengths rounded into the period), a masked reset leaves the other lanes alone, the periodic one repeats with its
    period, `grad` is the derivative of the value."""
    from jiminy_amd.processes import PeriodicPerlinProcess, RandomPerlinProcess
    from oracle import terrain_numpy
    B = 6
    times = (0.0, 0.13, 1.7, -2.4, 31.9)
    rnd = RandomPerlinProcess(0.4, 5, B)
    rnd.reset(100)
    seeds = [5, 9, 11, 12, 400, 2 ** 40 + 3]
    per = PeriodicPerlinProcess(0.4, 3.0, 4, B)
    per.reset(torch.tensor(seeds))
    for t in times:
        got_r, got_p = rnd(t).numpy(), per(t).numpy()
        for lane in range(B):
            assert got_r[lane] == terrain_numpy.RandomPerlinProcess(0.4, 5, 1, 100 + lane)([t])
            assert got_p[lane] == terrain_numpy.PeriodicPerlinProcess(0.4, 3.0, 4, 1, seeds[lane])([t])
    # one time per lane
    tt = torch.linspace(-1.0, 2.0, B, dtype=torch.float64)
    got = per(tt).numpy()
    for lane in range(B):
        assert got[lane] == terrain_numpy.PeriodicPerlinProcess(0.4, 3.0, 4, 1, seeds[lane])([float(tt[lane])])
    assert float((per(0.37) - per(3.37)).abs().max()) < 1e-12 and float((per(0.37) - per(-2.63)).abs().max()) < 1e-12
    assert float((rnd(0.37) - rnd(3.37)).abs().min()) > 1e-6            # ... and the random one does not repeat
    # masked reset
    before = rnd(0.5).clone()
    mask = torch.tensor([1, 0, 0, 1, 0, 0], dtype=torch.uint8)
    rnd.reset(7000, lane_mask=mask)
    after = rnd(0.5)
    assert torch.equal(after[mask == 0], before[mask == 0]) and bool((after[mask == 1] != before[mask == 1]).all())
    assert float(after[3]) == terrain_numpy.RandomPerlinProcess(0.4, 5, 1, 7003)([0.5])
    # derivative, range
    h = 1e-6
    for p in (rnd, per):
        assert float((p.grad(0.77) - (p(0.77 + h) - p(0.77 - h)) / (2 * h)).abs().max()) < 1e-8
    dense = torch.stack([per(float(t)) for t in np.linspace(0.0, 3.0, 400)])
    assert 0.05 < float(dense.abs().max()) < 1.5
    with pytest.raises(ValueError):
        PeriodicPerlinProcess(0.4, 0.3, 2, B)
    # period / wavelength = 2.5 exactly: std::round goes AWAY from zero (3 knots, random.hxx:499-500), Python's round to even (2)
    half = PeriodicPerlinProcess(2.0, 5.0, 1, 2)
    half.reset(torch.tensor([5, 9]))
    assert half._size == [3] and half._octaves[0][0] == 5.0 / 3.0
    assert float(half(0.7)[0]) == terrain_numpy.PeriodicPerlinProcess(2.0, 5.0, 1, 1, 5)([0.7])
    with pytest.raises(ValueError):
        RandomPerlinProcess(0.4, 0, B)
