"""Fit of gelu(x) = x * (0.5 + w * Q(w^2)), w = clamp(x, -L, L)   [erf GELU: 0.5 + w Q(w^2) ~= Phi(x)]
used by gelu_erf_f / geglu_f (csrc/common.h): Lawson-reweighted least squares on Chebyshev nodes for the ABSOLUTE error of
gelu itself, with the linear constraint L * Q(L^2) = 0.5 (so that gelu(x) = x exactly for x >= L and 0 for x <= -L),
coefficients rounded to fp32, error of the fp32-evaluated form against scipy."""
import sys

import numpy as np
from scipy.special import erf

L = float(sys.argv[2]) if len(sys.argv) > 2 else 4.25
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = 8000
k = np.arange(N)
x = (np.cos(np.pi * (k + 0.5) / N) + 1) / 2 * L            # [0, L] (the error is even in x)
u = x * x
target = 0.5 * x * erf(x / np.sqrt(2))                      # gelu(x) - 0.5 x
# Q(u) = sum c_k u^k with sum c_k L^(2k) = 0.5 / L  ->  eliminate c_0
pw = np.vander(u, DEG + 1, increasing=True)                 # u^k
Lp = np.array([L ** (2 * j) for j in range(DEG + 1)])
A = (pw[:, 1:] - Lp[1:][None, :]) * (x * x)[:, None]         # basis after elimination (model = x * w * Q, w = x here)
b = target - (0.5 / L) * x * x
w = np.ones(N)
for _ in range(300):
    c, *_ = np.linalg.lstsq(A * w[:, None], b * w, rcond=None)
    err = np.abs(A @ c - b)
    w = w * (1 + 2 * err / err.max())
    w /= w.mean()
coef = np.concatenate([[0.5 / L - (c * Lp[1:]).sum()], c])
c32 = coef.astype(np.float32)
print("L = %.9g, degree %d" % (L, DEG))
print("Q coefficients q_k:", ", ".join("%.9ef" % v for v in c32))
xx = np.linspace(-12, 12, 4000001).astype(np.float32)
ww = np.clip(xx, np.float32(-L), np.float32(L)).astype(np.float32)
uu = (ww * ww).astype(np.float32)
q = np.float32(c32[-1]) * np.ones_like(uu)
for v in c32[-2::-1]:
    q = (q * uu + np.float32(v)).astype(np.float32)
r = (ww * q + np.float32(0.5)).astype(np.float32)
g = (xx * r).astype(np.float32)
ref = 0.5 * xx.astype(np.float64) * (1 + erf(xx.astype(np.float64) / np.sqrt(2)))
e = np.abs(g - ref)
print("max |gelu error| %.3e at x = %.3f;  beyond +-L: %.3e" % (e.max(), xx[e.argmax()], e[np.abs(xx) > L].max()))
