"""Fit of the sigmoid-quintic GELU used by the bf16 MFMA epilogues (whisperseg_amd/csrc/wseg_common.h: gelu_sig5).

    gelu(x) ~= x * sigmoid(x * (a1 + a3 x^2 + a5 x^4)),   minimax over [-9, 9] against 0.5 x (1 + erf(x / sqrt 2)).
Prints the coefficients, the same coefficients with -log2(e) folded in (what the kernel uses) and the max abs
error evaluated in float32 arithmetic.
"""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

x = np.linspace(-9, 9, 200001)
g = 0.5 * x * (1 + erf(x / np.sqrt(2)))


def approx(c, x=x):
    x2 = x * x
    with np.errstate(over="ignore"):
        return x / (1 + np.exp(-x * (c[0] + x2 * (c[1] + x2 * c[2]))))


best = None
for c0 in ([1.5957691, 0.0713548, 0.0], [1.5976, 0.07056, 0.0], [1.596, 0.0714, -1e-4], [1.5958, 0.073, -3e-4]):
    r = minimize(lambda c: np.max(np.abs(approx(c) - g)), c0, method="Nelder-Mead",
                 options=dict(xatol=1e-10, fatol=1e-12, maxiter=20000))
    if best is None or r.fun < best.fun:
        best = r
print("a1, a3, a5 =", best.x, " max |err| (float64) =", best.fun)
q = (-np.log2(np.e) * best.x).astype(np.float32)
print("kernel coefficients:", [repr(float(v)) for v in q])
xf = np.linspace(-12, 12, 400001).astype(np.float32)
x2 = np.minimum(xf * xf, np.float32(64))
qq = (x2 * q[2] + q[1]).astype(np.float32)
qq = (qq * x2 + q[0]).astype(np.float32)
with np.errstate(over="ignore"):
    e = np.exp2((xf * qq).astype(np.float32)).astype(np.float32)
y = (xf * (np.float32(1) / (np.float32(1) + e))).astype(np.float32)
gf = 0.5 * xf.astype(np.float64) * (1 + erf(xf.astype(np.float64) / np.sqrt(2)))
print("max |err| (float32 evaluation, [-12, 12]) =", np.max(np.abs(y - gf)))
