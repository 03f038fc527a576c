"""Fit of the odd polynomial erf(z) ~= z * P(z^2) on [0, 3] used by gelu_erf_f (csrc/common.h): Lawson-reweighted least
squares on Chebyshev nodes, coefficients rounded to fp32, error of the fp32-evaluated GELU against scipy."""
import numpy as np
from scipy.special import erf

Z, DEG, N = 3.0, 8, 6000
k = np.arange(N)
u = (np.cos(np.pi * (k + 0.5) / N) + 1) / 2 * Z * Z
z = np.sqrt(u)
w = np.ones(N)
for _ in range(200):
    A = np.vander(u, DEG + 1, increasing=True) * z[:, None]
    coef, *_ = np.linalg.lstsq(A * w[:, None], erf(z) * w, rcond=None)
    err = np.abs(A @ coef - erf(z))
    w = w * (1 + 2 * err / err.max())
    w /= w.mean()
c32 = coef.astype(np.float32)
print("coefficients c_k of erf(z) = z * sum c_k z^(2k):", ", ".join("%.9ef" % c for c in c32))
x = np.linspace(-8, 8, 2000001).astype(np.float32)
zz = np.minimum(np.abs(x) * np.float32(0.70710678), np.float32(Z)).astype(np.float32)
uu = (zz * zz).astype(np.float32)
p = np.float32(c32[-1]) * np.ones_like(uu)
for c in c32[-2::-1]:
    p = (p * uu + np.float32(c)).astype(np.float32)
g = (np.float32(0.5) * (x + np.abs(x) * (zz * p))).astype(np.float32)
ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
print("max |erf error| %.3e   max |gelu error| %.3e at x = %.3f" % (
    np.abs(zz * p - erf(zz.astype(np.float64))).max(), np.abs(g - ref).max(), x[np.abs(g - ref).argmax()]))
