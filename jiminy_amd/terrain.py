"""Random terrain generators on the device: `random_tile_ground` ≙ `jiminy.random_tile_ground` / `tiles`
(reference core/src/utilities/random.cc:488-656; core/include/jiminy/core/utilities/random.h:588).

The reference's generator is a pure function of the position: every tile (i, j) of a rotated, offset grid gets the
height `heightMax * u(i, j)`, `u` = XXH32 of the two int32 tile indices with the seed (random.cc:200-256, 488-501: kept
only when `hash % sparsity == 0`, else 0), and a band of width `interpDelta` on either side of a tile edge blends
linearly into the neighbouring tile.  Here the same function is evaluated for whole tensors of positions with integer
tensor arithmetic (wrap-around uint32 products carried in int64), on whatever device the positions live on, e.g. to
fill the height map of `BatchedEngine.set_ground_profile` from a seed without leaving the GPU.
"""
from __future__ import annotations

import math
from typing import Optional, Callable, Sequence, Tuple

import torch

_P1, _P2, _P3, _P4, _P5 = 2654435761, 2246822519, 3266489917, 668265263, 374761393
_M32 = 0xFFFFFFFF


def _u32(x: torch.Tensor) -> torch.Tensor:
    return x & _M32


def _rotl(x: torch.Tensor, r: int) -> torch.Tensor:
    return _u32((x << r) | (x >> (32 - r)))


def xxh32_words(words: Sequence[torch.Tensor], seed: int) -> torch.Tensor:
    """XXH32 (random.cc:200-256) of a short key given as 32-bit little-endian words (fewer than four: the
    `len < 16` branch), one hash per tensor element; int64 tensors holding values in [0, 2^32)."""
    n = 4 * len(words)
    assert n < 16
    h = torch.full_like(words[0], (int(seed) + _P5 + n) & _M32)
    for w in words:
        h = _u32(h + _u32(_u32(w) * _P3))
        h = _u32(_rotl(h, 17) * _P4)
    h = h ^ (h >> 15)
    h = _u32(h * _P2)
    h = h ^ (h >> 13)
    h = _u32(h * _P3)
    h = h ^ (h >> 16)
    return h


def _uniform_sparse(ix: torch.Tensor, iy: torch.Tensor, sparsity: int, seed: int) -> torch.Tensor:
    """`uniformSparseFromState(Vector2<int32>{ix, iy}, sparsity, seed)` (random.cc:488-509): float32 in [0, 1]."""
    h = xxh32_words([ix & _M32, iy & _M32], seed)
    # float(hash) / float(UINT32_MAX): the divisor rounds to 2^32 in float32, so the quotient is an exact scaling --
    # written as one, it is bit-identical on every device (a float32 division is not correctly rounded everywhere)
    u = h.to(torch.float32) * (1.0 / 4294967296.0)
    return torch.where(h % int(sparsity) == 0, u, torch.zeros_like(u))


def random_tile_ground(size: Tuple[float, float], height_max: float, interp_delta: Tuple[float, float],
                       sparsity: int, orientation: float, seed: int) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    """≙ `tiles(size, heightMax, interpDelta, sparsity, orientation, seed)` (random.cc:552-656): returns
    `heightmap(x, y) -> height` for float64 tensors `x`, `y` of any (equal) shape and device."""
    sx, sy = float(size[0]), float(size[1])
    thr = [min(max(float(interp_delta[0]), 0.01), sx / 2.0) / sx, min(max(float(interp_delta[1]), 0.01), sy / 2.0) / sy]
    sizes = (sx, sy)
    c, s = math.cos(float(orientation)), math.sin(float(orientation))

    def offset(i: int) -> float:   # size[i] * uniformSparseFromState(Vector1<Eigen::Index>{i}, 1, seed): an int64 key
        k = torch.tensor([i], dtype=torch.int64)
        h = xxh32_words([k, torch.zeros_like(k)], seed)
        return sizes[i] * float(h.to(torch.float32) * (1.0 / 4294967296.0))
    off = (offset(0), offset(1))

    def z_of(ix: torch.Tensor, iy: torch.Tensor) -> torch.Tensor:
        return float(height_max) * _uniform_sparse(ix, iy, sparsity, seed).to(torch.float64)

    def interp1d(ix, iy, rel, dim):
        """`tile2dInterp1d` (random.cc:511-550): height blended along `dim` where the point lies in an edge band."""
        z = z_of(ix, iy)
        dm = (-1, 0) if dim == 0 else (0, -1)
        z_m = z_of(ix + dm[0], iy + dm[1])
        z_p = z_of(ix - dm[0], iy - dm[1])
        t = thr[dim]
        lo, hi = rel < t, (1.0 - rel) < t
        h_lo = z + (z_m - z) * ((1.0 - rel / t) / 2.0)
        h_hi = z + (z_p - z) * ((1.0 + (rel - 1.0) / t) / 2.0)
        return torch.where(lo, h_lo, torch.where(hi, h_hi, z))

    def heightmap(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        x = torch.as_tensor(x, dtype=torch.float64)
        y = torch.as_tensor(y, dtype=torch.float64, device=x.device)
        px, py = x + off[0], y + off[1]
        rx, ry = (c * px - s * py) / sx, (s * px + c * py) / sy
        fx, fy = torch.floor(rx), torch.floor(ry)
        ix, iy = fx.to(torch.int64), fy.to(torch.int64)      # (int32 in the reference: same values for |index| < 2^31)
        rx, ry = rx - fx, ry - fy
        ex = (rx < thr[0]) | ((1.0 - rx) < thr[0])
        ey = (ry < thr[1]) | ((1.0 - ry) < thr[1])
        h_x = interp1d(ix, iy, rx, 0)            # edge along x only (or no edge at all: the tile's own height)
        h_y = interp1d(ix, iy, ry, 1)            # edge along y only
        # corner: the x-blended heights of this row of tiles and of the neighbouring row, blended along y
        t = thr[1]
        lo = ry < t
        h_0 = h_x
        h_n = torch.where(lo, interp1d(ix, iy - 1, rx, 0), interp1d(ix, iy + 1, rx, 0))
        ratio = torch.where(lo, (1.0 - ry / t) / 2.0, (1.0 + (ry - 1.0) / t) / 2.0)
        h_xy = h_0 + (h_n - h_0) * ratio
        return torch.where(ex & ey, h_xy, torch.where(ey, h_y, h_x))

    return heightmap


def periodic_stairs(step_width: float, step_height: float, step_number: int, orientation: float
                    ) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    """≙ `periodicStairs(stepWidth, stepHeight, stepNumber, orientation)` (core/src/utilities/geometry.cc:797-868):
    alternating ascending / descending staircases of `step_number` steps along the direction `orientation`, the vertical
    edge of every step replaced by a ramp over the last 1 % of the step (`interpDelta = 0.01`).  `heightmap(x, y) -> height`
    on float64 tensors of any shape and device."""
    interp_delta = 0.01
    ax, ay = math.cos(float(orientation)), math.sin(float(orientation))
    w, hgt, n = float(step_width), float(step_height), int(step_number)
    eps32 = float(torch.finfo(torch.float32).eps)

    def heightmap(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        x = torch.as_tensor(x, dtype=torch.float64)
        y = torch.as_tensor(y, dtype=torch.float64, device=x.device)
        pos = ax * x + ay * y
        mod = torch.fmod(pos.abs(), w * n * 2)
        idx = torch.floor(mod / w).to(torch.int64)                       # static_cast<uint32_t>(modPos / stepWidth)
        down = idx >= n
        idx = torch.where(down, 2 * n - idx, idx)
        sign = torch.where(down, -1.0, 1.0).to(torch.float64)
        height = idx.to(torch.float64) * hgt
        rel = torch.fmod(mod + eps32, w) / w
        slope = sign * hgt / interp_delta
        return torch.where((1.0 - rel) < interp_delta, height + slope * (rel - (1.0 - interp_delta)), height)
    return heightmap


def sum_heightmaps(heightmaps: Sequence[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]]
                   ) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    """≙ `sumHeightmaps` (geometry.cc:694-736): the heights add up (the normals of the reference are re-derived from the
    sampled height map by `BatchedEngine.set_ground_profile`, like for every ground here)."""
    if not heightmaps:
        raise ValueError("At least one heightmap must be specified.")
    if len(heightmaps) == 1:
        return heightmaps[0]
    return lambda x, y: sum(h(x, y) for h in heightmaps[1:]) + heightmaps[0](x, y)


def merge_heightmaps(heightmaps: Sequence[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]]
                     ) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    """≙ `mergeHeightmaps` (geometry.cc:738-795): the highest of the grounds at every point."""
    if not heightmaps:
        raise ValueError("At least one heightmap must be specified.")
    if len(heightmaps) == 1:
        return heightmaps[0]

    def heightmap(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        out = heightmaps[0](x, y)
        for h in heightmaps[1:]:
            out = torch.maximum(out, h(x, y))
        return out
    return heightmap


# ---- Perlin grounds (`randomPerlinGround`, `unidirectionalRandomPerlinGround`: core/src/utilities/geometry.cc:858-926 on
# `RandomPerlinProcess<N>`, core/include/jiminy/core/utilities/random.hxx:200-420, 564-690)
PERLIN_NOISE_PERSISTENCE, PERLIN_NOISE_LACUNARITY = 1.50, 0.85     # random.hxx:11-12


class _Pcg32:
    """Host copy of the reference's `PCG32` (random.cc:10-37): the two or three draws per octave that place a ground."""

    def __init__(self, seed: int) -> None:
        self.state = (int(seed) | 3) & 0xFFFFFFFFFFFFFFFF

    def __call__(self) -> int:
        self.state = (self.state * 6364136223846793005) & 0xFFFFFFFFFFFFFFFF
        s = self.state
        rshift = (s >> 61) & 7
        s ^= s >> 22
        return (s >> (22 + rshift)) & 0xFFFFFFFF

    def uniform(self) -> float:      # std::generate_canonical<float, 24>: one word
        r = torch.tensor(self(), dtype=torch.int64).to(torch.float32) * (1.0 / 4294967296.0)
        return min(float(r), float(torch.nextafter(torch.tensor(1.0), torch.tensor(0.0))))


def _fade(d: torch.Tensor) -> torch.Tensor:
    return d * d * d * (d * (d * 6.0 - 15.0) + 10.0)


def _perlin_process(wavelength: float, num_octaves: int, n: int, seed: int, period: Optional[float] = None):
    """`RandomPerlinProcess<n>(wavelength, numOctaves)` -- or, with `period`, `PeriodicPerlinProcess<n>(wavelength, period,
    numOctaves)` (random.hxx:491-556, 640-690: gradients from a table drawn once per octave, knots wrapped into the period) --
    reset with `PCG32(seed)`: returns f(list of n coordinate tensors)."""
    if num_octaves < 1:
        raise ValueError("'numOctaves' must at least 1.")
    if wavelength <= 0.0:
        raise ValueError("'wavelength' must be strictly larger than 0.0.")
    if period is not None and period < max(wavelength, wavelength / PERLIN_NOISE_LACUNARITY ** (num_octaves - 1)):
        raise ValueError("'period' must be larger than the wavelength of all the octaves")
    g = _Pcg32(seed)
    octaves, scale, wl = [], 1.0, float(wavelength)
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)  # noqa: E731
    for _ in range(int(num_octaves)):
        if period is None:
            shift = [g.uniform() for _ in range(n)]
            octaves.append((wl, scale, shift, g()))
        else:
            wl_p = float(period) / max(math.floor(float(period) / wl + 0.5), 1.0)     # std::round: half away from zero
            size = int(float(period) / wl_p)
            shift = [g.uniform() for _ in range(n)]
            table = []
            for _ in range(size ** n):
                if n == 1:
                    table.append([float(f32(g.uniform()) * f32(2.0) + f32(-1.0))])     # uniform_real_distribution<float>(-1, 1)
                else:
                    theta = 2.0 * math.pi * g.uniform()
                    radius = float(torch.sqrt(f32(g.uniform())))
                    table.append([radius * math.cos(theta), radius * math.sin(theta)])
            octaves.append((wl_p, scale, shift, (size, torch.tensor(table, dtype=torch.float64))))
        wl /= PERLIN_NOISE_LACUNARITY
        scale *= PERLIN_NOISE_PERSISTENCE
    amplitude = math.sqrt(sum(s * s for _, s, _, _ in octaves))
    fmax = 4294967295.0

    def grad_knot(knot: Sequence[torch.Tensor], oseed) -> Sequence[torch.Tensor]:
        if period is not None:
            size, table = oseed
            index, mul = torch.zeros_like(knot[0]), 1
            for k in knot:
                index = index + torch.remainder(k, size) * mul
                mul *= size
            gk = table.to(index.device)[index]
            return [gk[..., i] for i in range(n)]
        # xxHash of the int32 knot coordinates (4 bytes each): 1-D = one word, 2-D = two words
        h = xxh32_words([k.to(torch.int64) & 0xFFFFFFFF for k in knot], oseed)
        if n == 1:
            return [2.0 * (h.to(torch.float32) / torch.tensor(fmax, dtype=torch.float32)).to(torch.float64) - 1.0]
        # rejection sampling on the disk (random.hxx:438-452): every point keeps hashing until it is inside
        one, two, f = torch.tensor(1.0, dtype=torch.float32), torch.tensor(2.0, dtype=torch.float32), torch.tensor(fmax, dtype=torch.float32)
        gx = torch.zeros(h.shape, dtype=torch.float32, device=h.device)
        gy = torch.zeros_like(gx)
        todo = torch.ones(h.shape, dtype=torch.bool, device=h.device)
        for _ in range(64):
            x = two * h.to(torch.float32) / f - one
            h = xxh32_words([h], oseed)
            y = two * h.to(torch.float32) / f - one
            ok = todo & (x * x + y * y <= one)
            gx, gy = torch.where(ok, x, gx), torch.where(ok, y, gy)
            todo = todo & ~ok
            if not bool(todo.any()):
                break
        return [gx.to(torch.float64), gy.to(torch.float64)]

    def octave(coords: Sequence[torch.Tensor], wl_: float, shift: Sequence[float], oseed) -> torch.Tensor:
        cell = [coords[i] / wl_ + shift[i] for i in range(n)]
        left = [torch.floor(c) for c in cell]
        dl = [cell[i] - left[i] for i in range(n)]
        dr = [d - 1.0 for d in dl]
        li = [l_.to(torch.int64) for l_ in left]
        offsets = []
        for k in range(1 << n):
            knot = [li[i] + 1 if k & (1 << i) else li[i] for i in range(n)]
            delta = [dr[i] if k & (1 << i) else dl[i] for i in range(n)]
            gk = grad_knot(knot, oseed)
            offsets.append(sum(gk[i] * delta[i] for i in range(n)))
        ratio = [_fade(d) for d in dl]
        for i in range(n - 1, -1, -1):
            for k in range(1 << i):
                offsets[k] = offsets[k] + ratio[i] * (offsets[k | (1 << i)] - offsets[k])
        return offsets[0]

    def process(coords: Sequence[torch.Tensor]) -> torch.Tensor:
        return sum(s * octave(coords, wl_, sh, sd) for wl_, s, sh, sd in octaves) / amplitude
    return process


def random_perlin_ground(wavelength: float, num_octaves: int, seed: int) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    """≙ `randomPerlinGround(wavelength, numOctaves, seed)` (geometry.cc:921-926): 2-D gradient noise in [-1, 1], octaves of
    wavelength / 0.85^i weighted 1.5^i; `heightmap(x, y) -> height` on float64 tensors."""
    fun = _perlin_process(wavelength, num_octaves, 2, seed)

    def heightmap(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        x = torch.as_tensor(x, dtype=torch.float64)
        return fun([x, torch.as_tensor(y, dtype=torch.float64, device=x.device)])
    return heightmap


def unidirectional_random_perlin_ground(wavelength: float, num_octaves: int, orientation: float, seed: int
                                        ) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    """≙ `unidirectionalRandomPerlinGround(wavelength, numOctaves, orientation, seed)` (geometry.cc:913-919)."""
    fun = _perlin_process(wavelength, num_octaves, 1, seed)
    ax, ay = math.cos(float(orientation)), math.sin(float(orientation))

    def heightmap(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        x = torch.as_tensor(x, dtype=torch.float64)
        return fun([ax * x + ay * torch.as_tensor(y, dtype=torch.float64, device=x.device)])
    return heightmap


def periodic_perlin_ground(wavelength: float, period: float, num_octaves: int, seed: int
                           ) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    """≙ `periodicPerlinGround(wavelength, period, numOctaves, seed)` (geometry.cc:928-934): the same noise, periodic in x and y."""
    fun = _perlin_process(wavelength, num_octaves, 2, seed, period=period)

    def heightmap(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        x = torch.as_tensor(x, dtype=torch.float64)
        return fun([x, torch.as_tensor(y, dtype=torch.float64, device=x.device)])
    return heightmap


def unidirectional_periodic_perlin_ground(wavelength: float, period: float, num_octaves: int, orientation: float, seed: int
                                          ) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    """≙ `unidirectionalPeriodicPerlinGround(wavelength, period, numOctaves, orientation, seed)` (geometry.cc:936-945)."""
    fun = _perlin_process(wavelength, num_octaves, 1, seed, period=period)
    ax, ay = math.cos(float(orientation)), math.sin(float(orientation))

    def heightmap(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        x = torch.as_tensor(x, dtype=torch.float64)
        return fun([ax * x + ay * torch.as_tensor(y, dtype=torch.float64, device=x.device)])
    return heightmap
