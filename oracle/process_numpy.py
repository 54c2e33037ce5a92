"""Scalar restatement of the reference's periodic Gaussian process -- TEST INFRASTRUCTURE ONLY.

Follows `PeriodicTabularProcess` / `PeriodicGaussianProcess` (core/include/jiminy/core/utilities/random.h:317-379,
core/src/utilities/random.cc:322-458) and `internal::standardToeplitzCholeskyLower` (random.hxx:159-189) one
statement at a time, for ONE realisation, with dense loops instead of the product's batched tensor operations.
"""
import math

import numpy as np


def standard_toeplitz_cholesky_lower(coeffs, reg):
    n = len(coeffs)
    l = np.zeros((n, n))
    g = np.zeros((2, n))
    for j in range(n):
        g[0, j] = coeffs[j]
        g[1, j] = coeffs[j]
    g[0, 0] += reg
    for j in range(n):
        l[j, 0] = g[0, j]
    row0 = g[0].copy()
    for j in range(1, n):
        g[0, j] = row0[j - 1]
    for i in range(1, n):
        rho = -g[1, i] / g[0, i]
        s = math.sqrt((1.0 - rho) * (1.0 + rho))
        for j in range(i, n):
            a, b = g[0, j], g[1, j]
            g[0, j] = (a + rho * b) / s
            g[1, j] = (rho * a + b) / s
        for j in range(i, n):
            l[j, i] = g[0, j]
        row0 = g[0].copy()
        for j in range(i + 1, n):
            g[0, j] = row0[j - 1]
    return l


class PeriodicGaussianProcess:
    def __init__(self, wavelength, period):
        self.wavelength, self.period = wavelength, period
        self.num_times = int(math.ceil(period / (0.1 * wavelength)))
        self.dt = period / float(self.num_times)
        n = self.num_times
        coeffs = [math.exp(-2.0 * (math.sin(math.pi / n * i) / wavelength) ** 2) for i in range(n)]
        self.cov_sqrt_root = standard_toeplitz_cholesky_lower(coeffs, 1e-9)
        self.cov_jacobian = np.zeros((n, n))
        for i in range(n):
            for j in range(n):
                self.cov_jacobian[i, j] = (-2 * math.pi / period / wavelength ** 2 * math.sin(2 * math.pi / n * (i - j))
                                           * math.exp(-2.0 * (math.sin(math.pi / n * (i - j)) / wavelength) ** 2))
        self.values = np.zeros(n)
        self.grads = np.zeros(n)

    def reset(self, normal_vec):
        n = self.num_times
        L = np.tril(self.cov_sqrt_root)
        self.values = L @ normal_vec
        # back substitution of L^T x = z
        x = np.zeros(n)
        for i in range(n - 1, -1, -1):
            s = normal_vec[i]
            for j in range(i + 1, n):
                s -= L[j, i] * x[j]
            x[i] = s / L[i, i]
        self.grads = self.cov_jacobian @ x

    def _knots(self, t):
        period = float(self.num_times) * self.dt
        value = math.fmod(t, period)
        if value < 0.0:
            value += period
        quot = value / self.dt
        left = int(math.floor(quot))
        right = left + 1
        ratio = quot - float(left)
        if right == self.num_times:
            right = 0
        return left, right, ratio

    def __call__(self, t):
        il, ir, ratio = self._knots(t)
        dy = self.values[ir] - self.values[il]
        a = self.grads[il] * self.dt - dy
        b = -self.grads[ir] * self.dt + dy
        return self.values[il] + ratio * ((1.0 - ratio) * ((1.0 - ratio) * a + ratio * b) + dy)

    def grad(self, t):
        il, ir, ratio = self._knots(t)
        dy = self.values[ir] - self.values[il]
        a = self.grads[il] * self.dt - dy
        b = -self.grads[ir] * self.dt + dy
        return ((1.0 - ratio) * (1.0 - 3.0 * ratio) * a + ratio * (2.0 - 3.0 * ratio) * b + dy) / self.dt


class PeriodicFourierProcess(PeriodicGaussianProcess):
    """random.h:389-409 / random.cc:462-485, scalar, one realisation; evaluation inherited (PeriodicTabularProcess)."""

    def __init__(self, wavelength, period):
        self.wavelength, self.period = wavelength, period
        self.num_times = int(math.ceil(period / (0.1 * wavelength)))
        self.dt = period / float(self.num_times)
        self.num_harmonics = int(math.ceil(period / wavelength))
        n, H = self.num_times, self.num_harmonics
        self.cos_mat = np.array([[math.cos(2 * math.pi / n * i * (j + 1)) for j in range(H)] for i in range(n)])
        self.sin_mat = np.array([[math.sin(2 * math.pi / n * i * (j + 1)) for j in range(H)] for i in range(n)])
        self.values = np.zeros(n)
        self.grads = np.zeros(n)

    def reset(self, normal_vec1, normal_vec2):
        H = self.num_harmonics
        scale = math.sqrt(2.0) / math.sqrt(2 * H + 1)
        self.values = scale * self.sin_mat @ normal_vec1 + scale * self.cos_mat @ normal_vec2
        diff = 2 * math.pi / self.period * np.linspace(1.0, float(H), H)
        self.grads = scale * self.cos_mat @ (normal_vec1 * diff) - scale * self.sin_mat @ (normal_vec2 * diff)
