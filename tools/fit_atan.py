"""Fits the odd polynomial atan(t) ~ t + t s Q(s), s = t^2, t in [0, 1], used by angle_turns / angle_bins36 (keypoint_kernels.hip.h:
atan2 by the half-angle tangent, t = y / (|v| + |x|)), and reports the error of the float32 evaluation against the double-precision
atan2 over random arguments.
usage: python tools/fit_atan.py [n_terms] [pin]
  pin: the fit is constrained to be exact at t = tan(pi / 8) = sqrt(2) - 1 -- the argument of an exactly diagonal gradient
       (|dx| == |dy|), which sits on a boundary of the 36-bin orientation histogram (4.5 and 13.5 bins); also emulates the bin
       decision against the reference's f32 expression round(36 (atan2f / 2 pi)) on gradients on and next to the diagonal.
       NB the emulation's sqrt and division are correctly rounded; v_sqrt_f32 / v_rcp_f32 are not, and on the GPU exact diagonals scatter
       over both sides of the boundary whatever the bias -- which is why orientation_kernel sends the samples next to a boundary through
       the octant form instead (NOTEBOOK.md section 14); the pinned coefficients are the shipped ones."""
import sys
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pin = len(sys.argv) > 2 and sys.argv[2] == "pin"
# Chebyshev nodes in t, weighted least squares on (atan(t) - t) / t^3 = Q(t^2); a few Remez-like reweighting rounds
t = 0.5 * (1 - np.cos(np.pi * (np.arange(4000) + 0.5) / 4000))
t = t[t > 1e-3]
s = t * t
f = (np.arctan(t) - t) / (t * s)
t0 = np.sqrt(2.0) - 1.0
s0 = t0 * t0
f0 = (np.arctan(t0) - t0) / (t0 * s0)
wgt = np.ones_like(t)
V = np.vander(s, n, increasing=True)
V0 = np.vander(np.array([s0]), n, increasing=True)[0]
for it in range(60):
    if pin:          # eliminate c[0] through Q(s0) = f0
        A = (V[:, 1:] - V0[1:][None, :]) * (wgt * t * s)[:, None]
        ck, *_ = np.linalg.lstsq(A, (f - f0) * wgt * t * s, rcond=None)
        c = np.concatenate([[f0 - (ck * V0[1:]).sum()], ck])
    else:
        A = V * (wgt * t * s)[:, None]
        c, *_ = np.linalg.lstsq(A, f * wgt * t * s, rcond=None)
    err = np.abs((V @ c) * t * s + t - np.arctan(t))
    wgt = wgt * (1 + 0.5 * err / err.max())
print("max abs error of the double evaluation: %.3g rad (at tan(pi/8): %.3g)" % (err.max(), abs((V0 @ c) * t0 * s0 + t0 - np.arctan(t0))))
for name, k in (("radians", 1.0), ("turns: x 1/pi (angle_turns)", 1 / np.pi), ("10-degree bins: x 36/pi (angle_bins36)", 36 / np.pi)):
    print("coefficients, %s (ascending in s; leading term %.9e):" % (name, np.float32(k)))
    for v in c:
        print("    %.9ef," % np.float32(v * k))


def half_angle_f32(dx, dy, c32, lead):
    """angle_turns / angle_bins36 in float32: magnitude-signed angle of atan2(dx, dy) before the reflection for dy < 0."""
    m2 = (dx * dx + (dy * dy + np.float32(1e-30))).astype(np.float32)
    mag = np.sqrt(m2).astype(np.float32)
    tt = (dx * (np.float32(1) / (mag + np.abs(dy)).astype(np.float32))).astype(np.float32)
    ss = (tt * tt).astype(np.float32)
    q = np.full_like(ss, c32[-1])
    for v in c32[-2::-1]:
        q = (q * ss + v).astype(np.float32)
    return tt, (ss * q + np.float32(lead)).astype(np.float32)


rng = np.random.default_rng(1)
y = rng.standard_normal(2_000_000).astype(np.float32) * np.float32(0.2)
x = rng.standard_normal(2_000_000).astype(np.float32) * np.float32(0.2)
tt, p = half_angle_f32(y, x, (c / np.pi).astype(np.float32), 1 / np.pi)
r = (tt * p).astype(np.float32)
r = np.where(x < 0, np.float32(0.5) - r, r)
ref = np.arctan2(y.astype(np.float64), x.astype(np.float64)) / (2 * np.pi)
e = np.abs(r - ref)
e = np.minimum(e, 1 - e) * 2 * np.pi
print("float32 evaluation of the half-angle form: max abs err %.3g rad, mean %.3g" % (e.max(), e.mean()))

if pin:
    dy = rng.uniform(0.001, 0.5, 2_000_000).astype(np.float32)
    dx = (dy * (1 + rng.uniform(-3e-6, 3e-6, dy.size))).astype(np.float32)
    dx[:200000] = dy[:200000]                                  # exact diagonals
    tt, p = half_angle_f32(dx, dy, (c * 36 / np.pi).astype(np.float32), 36 / np.pi)
    o = np.arctan2(dx, dy).astype(np.float32)                  # the reference's expression in f32 (SIFTOrientation.metal:122-129)
    bo = np.floor(((o / np.float32(2 * np.pi)).astype(np.float32) * np.float32(36)).astype(np.float32) + np.float32(0.5))
    for eps in (0.0, 1e-6, 2e-6, 1e-5):
        r = (tt * p - np.float32(eps)).astype(np.float32)      # fma(t, p, -eps)
        bm = np.floor(r + np.float32(0.5))
        print("bias %g: bins that differ from the reference's f32 expression: %d of 200000 exact diagonals, %d of 1800000 within 3e-6 of the diagonal"
              % (eps, (bm[:200000] != bo[:200000]).sum(), (bm[200000:] != bo[200000:]).sum()))
