"""Scalar restatement of the reference's random tile terrain. TEST INFRASTRUCTURE ONLY (same rules as oracle.cpp).

Follows core/src/utilities/random.cc line by line: `xxHash` (:200-256), `uniformSparseFromStateImpl` (:488-501),
`tile2dInterp1d` (:511-550), `tiles` (:552-656).  Pinned: the hash against the reference's own compiled text
(tests/test_reference_cpp_leaves.py) and, for keys shorter than 16 bytes -- where the reference IS XXH32 --, against the independent
`xxhash` package (tests/test_terrain.py); the generator on its structural laws (constant tile interiors,
continuity across the blend bands, sparsity)."""
import math
import struct

import numpy as np

P1, P2, P3, P4, P5 = 2654435761, 2246822519, 3266489917, 668265263, 374761393
M = 0xFFFFFFFF


def rotl32(x, r):
    return ((x << r) | (x >> (32 - r))) & M


def xx_hash(data: bytes, seed: int) -> int:
    n, i = len(data), 0
    if n >= 16:
        v1, v2, v3, v4 = (seed + P1 + P2) & M, (seed + P2) & M, seed & M, (seed - P1) & M
        rnd = lambda acc, w: (rotl32((acc + w * P2) & M, 13) * P1) & M  # noqa: E731
        while i <= n - 16:
            w = struct.unpack_from("<4I", data, i)
            v1, v2, v3, v4 = rnd(v1, w[0]), rnd(v2, w[1]), rnd(v3, w[2]), rnd(v4, w[3])
            i += 16
        h = (rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18)) & M
        # random.cc:228 `len &= 15;` BEFORE `hash += len` (:236): for keys of 16 bytes or more the reference adds the
        # length of the TAIL, where XXH32 adds the whole length -- found by the reference-compiled fixtures
        # (tests/golden/ref_cpp_leaves.npz, round 6).  The reference only hashes 4- and 8-byte keys on the hot path.
        n_added = n & 15
    else:
        h = (seed + P5) & M
        n_added = n
    h = (h + n_added) & M
    while n - i >= 4:
        h = (h + struct.unpack_from("<I", data, i)[0] * P3) & M
        h = (rotl32(h, 17) * P4) & M
        i += 4
    while i < n:
        h = (h + data[i] * P5) & M
        h = (rotl32(h, 11) * P1) & M
        i += 1
    h ^= h >> 15
    h = (h * P2) & M
    h ^= h >> 13
    h = (h * P3) & M
    h ^= h >> 16
    return h


def uniform_sparse(data: bytes, sparsity: int, seed: int) -> float:
    h = xx_hash(data, seed)
    if h % sparsity == 0:
        return float(np.float32(h) / np.float32(M))
    return 0.0


def tile_2d_interp_1d(idx, rel, dim, size, sparsity, height_max, thr, seed):
    key = lambda ij: struct.pack("<2i", *ij)  # noqa: E731
    idx = list(idx)
    z = height_max * uniform_sparse(key(idx), sparsity, seed)
    if rel[dim] < thr[dim]:
        idx[dim] -= 1
        z_m = height_max * uniform_sparse(key(idx), sparsity, seed)
        idx[dim] += 1
        ratio = (1.0 - rel[dim] / thr[dim]) / 2.0
        return z + (z_m - z) * ratio, (z - z_m) / (2.0 * size[dim] * thr[dim])
    if 1.0 - rel[dim] < thr[dim]:
        idx[dim] += 1
        z_p = height_max * uniform_sparse(key(idx), sparsity, seed)
        idx[dim] -= 1
        ratio = (1.0 + (rel[dim] - 1.0) / thr[dim]) / 2.0
        return z + (z_p - z) * ratio, (z_p - z) / (2.0 * size[dim] * thr[dim])
    return z, 0.0


def tiles(size, height_max, interp_delta, sparsity, orientation, seed):
    size = [float(size[0]), float(size[1])]
    thr = [min(max(float(interp_delta[i]), 0.01), size[i] / 2.0) / size[i] for i in range(2)]
    offset = [size[i] * uniform_sparse(struct.pack("<q", i), 1, seed) for i in range(2)]
    c, s = math.cos(orientation), math.sin(orientation)

    def heightmap(x, y):
        px, py = x + offset[0], y + offset[1]
        rel = [(c * px - s * py) / size[0], (s * px + c * py) / size[1]]
        idx = [int(math.floor(rel[0])), int(math.floor(rel[1]))]
        rel = [rel[0] - idx[0], rel[1] - idx[1]]
        edge = [rel[i] < thr[i] or 1.0 - rel[i] < thr[i] for i in range(2)]
        if edge[0] and not edge[1]:
            return tile_2d_interp_1d(idx, rel, 0, size, sparsity, height_max, thr, seed)[0]
        if not edge[0] and edge[1]:
            return tile_2d_interp_1d(idx, rel, 1, size, sparsity, height_max, thr, seed)[0]
        if edge[0] and edge[1]:
            h0, _ = tile_2d_interp_1d(idx, rel, 0, size, sparsity, height_max, thr, seed)
            if rel[1] < thr[1]:
                hm, _ = tile_2d_interp_1d([idx[0], idx[1] - 1], rel, 0, size, sparsity, height_max, thr, seed)
                return h0 + (hm - h0) * ((1.0 - rel[1] / thr[1]) / 2.0)
            hp, _ = tile_2d_interp_1d([idx[0], idx[1] + 1], rel, 0, size, sparsity, height_max, thr, seed)
            return h0 + (hp - h0) * ((1.0 + (rel[1] - 1.0) / thr[1]) / 2.0)
        return height_max * uniform_sparse(struct.pack("<2i", *idx), sparsity, seed)
    return heightmap


def periodic_stairs(step_width, step_height, step_number, orientation):
    """geometry.cc:797-868, statement by statement (height only), one point at a time."""
    interp_delta = 0.01
    axis = (math.cos(orientation), math.sin(orientation))

    def heightmap(x, y):
        pos_rel = axis[0] * x + axis[1] * y
        mod_pos = math.fmod(abs(pos_rel), step_width * step_number * 2)
        stair_index = int(mod_pos / step_width)
        sign = 1
        if stair_index >= step_number:
            stair_index = 2 * step_number - stair_index
            sign = -1
        height = stair_index * step_height
        pos_rel_on_step = math.fmod(mod_pos + 1.1920929e-07, step_width) / step_width
        if 1.0 - pos_rel_on_step < interp_delta:
            slope = sign * step_height / interp_delta
            height += slope * (pos_rel_on_step - (1.0 - interp_delta))
        return height
    return heightmap


def sum_heightmaps(heightmaps):
    return lambda x, y: sum(h(x, y) for h in heightmaps)


def merge_heightmaps(heightmaps):
    return lambda x, y: max(h(x, y) for h in heightmaps)


# ---- Perlin grounds: `RandomPerlinProcess<N>` (core/include/jiminy/core/utilities/random.hxx:200-420, 564-690) behind
# `randomPerlinGround` / `unidirectionalRandomPerlinGround` (core/src/utilities/geometry.cc:858-926), scalar, N = 1 or 2
PERLIN_NOISE_PERSISTENCE, PERLIN_NOISE_LACUNARITY = 1.50, 0.85     # random.hxx:11-12


class Pcg32:
    """`PCG32` of the reference (random.cc:10-37): 64-bit MCG, XSH-RS output; `uniform()` = generate_canonical<float, 24>."""

    def __init__(self, seed):
        self.state = (int(seed) | 3) & M64

    def __call__(self):
        self.state = (self.state * 6364136223846793005) & M64
        s = self.state
        rshift = (s >> 61) & 7
        s ^= s >> 22
        return (s >> (22 + rshift)) & M32

    def uniform(self):
        import numpy as np
        r = np.float32(self()) * np.float32(1.0 / 4294967296.0)
        return float(r) if r < np.float32(1.0) else float(np.nextafter(np.float32(1.0), np.float32(0.0)))


M64, M32 = (1 << 64) - 1, (1 << 32) - 1


def _fade(d):
    return d * d * d * (d * (d * 6.0 - 15.0) + 10.0)


class RandomPerlinOctave:
    def __init__(self, wavelength, n, g):
        import numpy as np
        self.wavelength, self.n = wavelength, n
        self.shift = [g.uniform() for _ in range(n)]       # AbstractPerlinNoiseOctave::reset: uniform(N, 1, g)
        self.seed = g()                                    # RandomPerlinNoiseOctave::reset: seed_ = g()
        self._np = np

    def grad_knot(self, knot):
        np = self._np
        fmax = np.float32(4294967295.0)
        h = xx_hash(struct.pack("<%di" % self.n, *knot), self.seed)
        if self.n == 1:
            return [2.0 * float(np.float32(h) / fmax) - 1.0]
        while True:                                        # rejection sampling on the disk (random.hxx:438-452)
            x = np.float32(2) * np.float32(h) / fmax - np.float32(1)
            h = xx_hash(struct.pack("<I", h), self.seed)
            y = np.float32(2) * np.float32(h) / fmax - np.float32(1)
            if x * x + y * y <= np.float32(1):
                return [float(x), float(y)]

    def __call__(self, x):
        n = self.n
        cell = [x[i] / self.wavelength + self.shift[i] for i in range(n)]
        left = [int(math.floor(c)) for c in cell]
        dl = [cell[i] - left[i] for i in range(n)]
        dr = [d - 1.0 for d in dl]
        offsets = []
        for k in range(1 << n):
            knot = [left[i] + 1 if k & (1 << i) else left[i] for i in range(n)]
            delta = [dr[i] if k & (1 << i) else dl[i] for i in range(n)]
            g = self.grad_knot(knot)
            offsets.append(sum(g[i] * delta[i] for i in range(n)))
        ratio = [_fade(d) for d in dl]
        for i in range(n - 1, -1, -1):
            for k in range(1 << i):
                offsets[k] = offsets[k] + ratio[i] * (offsets[k | (1 << i)] - offsets[k])
        return offsets[0]


class RandomPerlinProcess:
    def __init__(self, wavelength, num_octaves, n, seed):
        g = Pcg32(seed)
        self.octaves, scale = [], 1.0
        for _ in range(num_octaves):
            self.octaves.append([wavelength, scale, None])
            wavelength /= PERLIN_NOISE_LACUNARITY
            scale *= PERLIN_NOISE_PERSISTENCE
        self.amplitude = math.sqrt(sum(s * s for _, s, _ in self.octaves))
        for o in self.octaves:                              # AbstractPerlinProcess::reset: the octaves in order, one generator
            o[2] = RandomPerlinOctave(o[0], n, g)

    def __call__(self, x):
        return sum(s * o(x) for _, s, o in self.octaves) / self.amplitude


def random_perlin_ground(wavelength, num_octaves, seed):
    fun = RandomPerlinProcess(wavelength, num_octaves, 2, seed)
    return lambda x, y: fun([x, y])


def unidirectional_random_perlin_ground(wavelength, num_octaves, orientation, seed):
    fun = RandomPerlinProcess(wavelength, num_octaves, 1, seed)
    ax = (math.cos(orientation), math.sin(orientation))
    return lambda x, y: fun([ax[0] * x + ax[1] * y])


class PeriodicPerlinOctave(RandomPerlinOctave):
    """`PeriodicPerlinNoiseOctave<N>` (random.h:483-506, random.hxx:491-556): gradients from a table of size^N entries drawn
    once, knots wrapped into the period."""

    def __init__(self, wavelength, period, n, g):
        import numpy as np
        if period < wavelength:
            raise ValueError("'period' must be larger than 'wavelength'.")
        self.wavelength = period / max(math.floor(period / wavelength + 0.5), 1.0)    # std::round (random.hxx:499): half away from zero
        self.n, self.period, self._np = n, period, np
        self.size = int(period / self.wavelength)
        self.shift = [g.uniform() for _ in range(n)]
        self.grads = []
        for _ in range(self.size ** n):
            if n == 1:
                # std::uniform_real_distribution<float>(-1, 1): generate_canonical * (b - a) + a, in float
                self.grads.append([float(np.float32(g.uniform()) * np.float32(2.0) + np.float32(-1.0))])
            else:
                theta = 2 * math.pi * g.uniform()
                radius = float(np.sqrt(np.float32(g.uniform())))
                self.grads.append([radius * math.cos(theta), radius * math.sin(theta)])

    def grad_knot(self, knot):
        index, shift = 0, 1
        for i in range(self.n):
            index += (knot[i] % self.size) * shift
            shift *= self.size
        return self.grads[index]


class PeriodicPerlinProcess(RandomPerlinProcess):
    def __init__(self, wavelength, period, num_octaves, n, seed):
        final = wavelength / PERLIN_NOISE_LACUNARITY ** (num_octaves - 1)
        if period < max(wavelength, final):
            raise ValueError("'period' must be larger than the wavelength of all the octaves")
        g = Pcg32(seed)
        self.octaves, scale = [], 1.0
        for _ in range(num_octaves):
            self.octaves.append([wavelength, scale, None])
            wavelength /= PERLIN_NOISE_LACUNARITY
            scale *= PERLIN_NOISE_PERSISTENCE
        self.amplitude = math.sqrt(sum(s * s for _, s, _ in self.octaves))
        for o in self.octaves:
            o[2] = PeriodicPerlinOctave(o[0], period, n, g)


def periodic_perlin_ground(wavelength, period, num_octaves, seed):
    fun = PeriodicPerlinProcess(wavelength, period, num_octaves, 2, seed)
    return lambda x, y: fun([x, y])


def unidirectional_periodic_perlin_ground(wavelength, period, num_octaves, orientation, seed):
    fun = PeriodicPerlinProcess(wavelength, period, num_octaves, 1, seed)
    ax = (math.cos(orientation), math.sin(orientation))
    return lambda x, y: fun([ax[0] * x + ax[1] * y])
