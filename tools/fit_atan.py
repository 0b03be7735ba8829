"""Fits the odd polynomial atan(t) ~ t + t s Q(s), s = t^2, t in [0, 1], used by atan2_lean (keypoint_kernels.hip.h), and
reports the error of the float32 evaluation against the double-precision atan2 over random arguments.
usage: python tools/fit_atan.py [n_terms]"""
import sys
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
# Chebyshev nodes in t, weighted least squares on (atan(t) - t) / t^3 = Q(t^2); a few Remez-like reweighting rounds
t = 0.5 * (1 - np.cos(np.pi * (np.arange(4000) + 0.5) / 4000))
t = t[t > 1e-3]
s = t * t
f = (np.arctan(t) - t) / (t * s)
wgt = np.ones_like(t)
for it in range(40):
    A = np.vander(s, n, increasing=True) * (wgt * t * s)[:, None]
    c, *_ = np.linalg.lstsq(A, f * wgt * t * s, rcond=None)
    err = np.abs((np.vander(s, n, increasing=True) @ c) * t * s + t - np.arctan(t))
    wgt = wgt * (1 + 0.5 * err / err.max())
print("coefficients (ascending in s):")
for v in c:
    print("    %.9ef," % np.float32(v))
print("max abs error of the double evaluation: %.3g" % err.max())

c32 = c.astype(np.float32)
rng = np.random.default_rng(1)
y = rng.standard_normal(2_000_000).astype(np.float32) * np.float32(0.2)
x = rng.standard_normal(2_000_000).astype(np.float32) * np.float32(0.2)
ax, ay = np.abs(x), np.abs(y)
mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
tt = (mn * (np.float32(1) / np.maximum(mx, np.float32(1e-30)))).astype(np.float32)
ss = tt * tt
q = np.full_like(ss, c32[-1])
for v in c32[-2::-1]:
    q = q * ss + v
r = tt + tt * (ss * q)
r = np.where(ay > ax, np.float32(np.pi / 2) - r, r)
r = np.where(x < 0, np.float32(np.pi) - r, r)
r = np.copysign(r, y).astype(np.float32)
ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
e = np.abs(r - ref)
ulp = np.spacing(np.abs(ref).astype(np.float32))
print("float32 evaluation: max abs err %.3g rad, max %.2f ulp, mean %.3f ulp" % (e.max(), (e / ulp).max(), (e / ulp).mean()))
e2 = np.abs(np.arctan2(y, x).astype(np.float32) - ref)
print("numpy float32 arctan2 for comparison: max %.2f ulp" % (e2 / ulp).max())
