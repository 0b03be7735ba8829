"""A second, independent statement of the orientation and descriptor stages: a literal numpy-float32 transcription of the
reference's Metal kernels, written from the .metal text without looking at oracle/sift_oracle.c, and cross-read against the
reference's commented-out CPU path (SIFT/SIFTCPU.swift:486-781, SIFT/SIFTPatch.swift:29-72), which states the same loops a
second time.  TEST INFRASTRUCTURE (like oracle/): used by tests/test_oracle_transcription.py to pin the oracle's two
least-pinned stages (they rested on one reading).  DESIGN.md section 3 holds the three-column concordance
(metal line <-> SIFTCPU line <-> oracle line) and the places where SIFTCPU.swift differs from the shipped Metal path.

Every arithmetic step is one float32 numpy operation in the order of the source expression (numpy does not contract or
reassociate), accumulations into the histograms go through np.add.at (unbuffered: one float32 add per element, in order), and
the sample loops are flattened in the source's loop order.  What may differ from the C oracle is libm only: numpy's float32
arctan2 / exp / cos / sin / power against glibc's (both within an ulp or two).

Source lines (Sources/MetalShaders/Metal/...):
  gradient            SIFTGradient.metal:15-39, Common.hpp:15-22
  orientation stage   SIFTOrientation.metal:16-175; host border filter + Int32 truncation SIFT/SIFTOctave.swift:303-337
  descriptor stage    SIFTDescriptor.metal:15-237; host expansion SIFT/SIFTOctave.swift:396-424
"""
import numpy as np

F = np.float32
PI_F = F(3.14159265358979323846)          # M_PI_F
ORI_BINS, DESC_BINS, DESC_SIDE = 36, 8, 4


def _round_half_away(v):
    """Metal / C round(): halves away from zero.  Exact in float64 (a float32 plus 0.5 is representable)."""
    v = np.asarray(v, np.float64)
    return np.where(v >= 0, np.floor(v + 0.5), -np.floor(-v + 0.5))


def _symmetrized(i, l):
    """Common.hpp:15-22 (i >= -2l here, so C's remainder and Python's agree)."""
    ll = 2 * l
    i = (np.asarray(i) + ll) % ll
    return np.where(i > l - 1, ll - 1 - i, i)


def gradient(G):
    """SIFTGradient.metal:15-39 for one Gaussian layer G [h, w] float32 -> (orientation, magnitude) textures.
    NB the argument order atan2(tx, ty) (the file's own FIXME, :34)."""
    h, w = G.shape
    xs, ys = np.arange(w), np.arange(h)
    px, mx = _symmetrized(xs + 1, w), _symmetrized(xs - 1, w)
    py, my = _symmetrized(ys + 1, h), _symmetrized(ys - 1, h)
    tx = (G[:, px] - G[:, mx]) * F(0.5)
    ty = (G[py, :] - G[my, :]) * F(0.5)
    oa = np.arctan2(tx, ty).astype(F)
    om = np.sqrt(tx * tx + ty * ty).astype(F)
    return oa, om


def _read(tex, x, y):
    """texture.read(ushort2(x, y)): 0 outside the image.  x, y integer arrays that may be negative or too large -- the ushort
    cast of such a value lands far outside any texture this code sees, so the read returns 0 either way."""
    h, w = tex.shape
    ok = (x >= 0) & (y >= 0) & (x < w) & (y < h)
    out = np.zeros(x.shape, F)
    out[ok] = tex[y[ok], x[ok]]
    return out


# ------------------------------------------------------------------------------------------------ orientation
def passes_border_filter(abs_x, abs_y, sigma, delta, w, h, lam=F(1.5)):
    """SIFTOctave.swift:303-329 (float coordinates, r = ceil(3 lambda sigma))."""
    x, y = F(abs_x) / F(delta), F(abs_y) / F(delta)
    s = F(sigma) / F(delta)
    r = np.ceil(F(3) * lam * s)
    min_x = min_y = F(1)
    max_x, max_y = F(w - 2), F(h - 2)
    return not (np.floor(x - r) < min_x or np.ceil(x + r) > max_x or np.floor(y - r) < min_y or np.ceil(y + r) > max_y)


def orientation_histogram(ori_tex, mag_tex, abs_x, abs_y, sigma_kp, delta, lam=F(1.5)):
    """getOrientationsHistogram, SIFTOrientation.metal:87-136.  abs_x / abs_y: the keypoint's float absolute coordinate; the host
    hands the kernel Int32(absoluteCoordinate) (SIFTOctave.swift:333-334: truncation toward zero)."""
    absolute_x, absolute_y = int(np.trunc(F(abs_x))), int(np.trunc(F(abs_y)))
    delta = F(delta)
    x = int(_round_half_away(F(absolute_x) / delta))                 # :99-100
    y = int(_round_half_away(F(absolute_y) / delta))
    sigma = F(sigma_kp) / delta                                      # :101
    exponent_denominator = F(2.0) * lam * lam                        # :103
    r = int(np.ceil(F(3) * lam * sigma))                             # :105
    jj, ii = np.meshgrid(np.arange(-r, r + 1), np.arange(-r, r + 1), indexing="ij")   # j outer, i inner (:107-108)
    jj, ii = jj.ravel(), ii.ravel()
    u = ii.astype(F) / sigma                                         # :111-114
    v = jj.astype(F) / sigma
    r2 = u * u + v * v
    wgt = np.exp(-r2 / exponent_denominator).astype(F)
    orientation = _read(ori_tex, x + ii, y + jj)                     # :117-119
    magnitude = _read(mag_tex, x + ii, y + jj)
    t = orientation / (F(2) * PI_F)                                  # :122
    b = _round_half_away(t * F(ORI_BINS)).astype(np.int64)           # :123
    b = np.where(b < 0, b + ORI_BINS, b)                             # :124-129 (each test once)
    b = np.where(b >= ORI_BINS, b - ORI_BINS, b)
    m = (wgt * magnitude).astype(F)                                  # :131
    hist = np.zeros(ORI_BINS, F)
    np.add.at(hist, b, m)                                            # :133, in loop order
    return hist


def smooth_histogram(hist, iterations=6):
    """smoothHistogram, SIFTOrientation.metal:67-84 (called with 6, :165)."""
    h = hist.astype(F).copy()
    n = ORI_BINS
    idx = np.arange(n)
    for _ in range(iterations):
        temp = h.copy()
        h = ((temp[(idx - 1 + n) % n] + temp) + temp[(idx + 1) % n]) / F(3.0)      # (h0 + h1 + h2) / 3.0, left to right
    return h


def principal_orientations(hist, orientation_threshold=F(0.8)):
    """getPrincipalOrientations + interpolatePeak + orientationFromBin, SIFTOrientation.metal:16-64."""
    n = ORI_BINS
    maximum = F(-2147483648.0)
    for i in range(n):
        maximum = max(maximum, hist[i])
    threshold = F(orientation_threshold) * maximum
    out = []
    tau = F(2) * PI_F
    for i in range(n):
        hm, h0, hp = hist[(i - 1 + n) % n], hist[i], hist[(i + 1) % n]
        if h0 > threshold and h0 > hm and h0 > hp:
            offset = (hm - hp) / (F(2) * (hm + hp - F(2) * h0))      # interpolatePeak(h1 = hm, h2 = h0, h3 = hp), :31-33
            t = (F(i) + offset) / F(n)                               # orientationFromBin, :16-28
            orientation = t * tau
            if orientation < 0:
                orientation = orientation + tau
            if orientation >= tau:
                orientation = orientation - tau
            out.append(F(orientation))
    return out


def orientations_of_octave(gauss_layers, delta, keypoints):
    """The orientation stage for one octave: keypoints = records with fields scale, absX, absY, sigma (the oracle's dtype).
    Returns [(keypoint index, [theta ...]) for the keypoints that pass the host border filter], and the smoothed histograms."""
    h, w = gauss_layers[0].shape
    grads = {}
    out, hists = [], []
    for k, kp in enumerate(keypoints):
        if not passes_border_filter(kp["absX"], kp["absY"], kp["sigma"], delta, w, h):
            continue
        s = int(kp["scale"])
        if s not in grads:
            grads[s] = gradient(gauss_layers[s])
        hist = smooth_histogram(orientation_histogram(grads[s][0], grads[s][1], kp["absX"], kp["absY"], kp["sigma"], delta))
        out.append((k, principal_orientations(hist)))
        hists.append(hist)
    return out, hists


# ------------------------------------------------------------------------------------------------ descriptor
def _normalize(f):
    """normalizeFeatures, SIFTDescriptor.metal:15-29: sequential float32 sum of squares."""
    magnitude = F(0)
    for v in f:
        magnitude = F(magnitude + F(v * v))
    d = F(1.0) / np.sqrt(magnitude, dtype=F)
    return (f * d).astype(F)


def descriptor(ori_tex, mag_tex, abs_x, abs_y, scale, sub_scale, theta, delta, scales_per_octave=3):
    """siftDescriptors, SIFTDescriptor.metal:120-237, for one (keypoint, theta).  Returns (int features [128], float features
    after the second normalisation [128]).  scalesPerOctave is the literal 3 of SIFTOctave.swift:398."""
    absolute_x, absolute_y = int(np.trunc(F(abs_x))), int(np.trunc(F(abs_y)))      # SIFTOctave.swift:417-418
    delta, theta = F(delta), F(theta)
    px = F(absolute_x) / delta                                       # :140-141
    py = F(absolute_y) / delta
    d, bins = DESC_SIDE, DESC_BINS
    tau = F(2) * PI_F
    cos_t, sin_t = np.cos(theta, dtype=F), np.sin(theta, dtype=F)    # :155-156
    bins_per_radian = F(bins) / tau
    exponent_denominator = F(d * d) * F(0.5)
    interval = F(scale) + F(sub_scale)
    intervals = F(scales_per_octave)
    sigma = F(1.6)
    sc = sigma * np.power(F(2.0), interval / intervals, dtype=F)     # :162
    histogram_width = F(3.0) * sc                                    # :164
    radius = int(histogram_width * np.sqrt(F(2.0), dtype=F) * (F(d) + F(1.0)) * F(0.5) + F(0.5))   # :165, int conversion truncates

    jj, ii = np.meshgrid(np.arange(-radius, radius + 1), np.arange(-radius, radius + 1), indexing="ij")   # j outer, i inner (:194-195)
    jj, ii = jj.ravel(), ii.ravel()
    fj, fi = jj.astype(F), ii.astype(F)
    rx = (fj * cos_t - fi * sin_t) / histogram_width                 # :197-198
    ry = (fj * sin_t + fi * cos_t) / histogram_width
    bx = rx + F(d // 2) - F(0.5)                                     # :199-200
    by = ry + F(d // 2) - F(0.5)
    # ushort2(px + j, py + i): float sum, conversion toward zero (a sum in (-1, 0) is texel 0; at or below -1 it is no texel)
    sx, sy = (px + fj).astype(F), (py + fi).astype(F)
    tx, ty = np.trunc(sx).astype(np.int64), np.trunc(sy).astype(np.int64)
    g_r = _read(ori_tex, tx, ty)                                     # :202
    g_g = _read(mag_tex, tx, ty)
    orientation = (g_r - theta).astype(F)                            # :203
    magnitude = g_g
    for _ in range(4):                                               # the while loops of :205-210 (|g.r - theta| < 3 pi: two turns suffice)
        orientation = np.where(orientation < 0, orientation + tau, orientation).astype(F)
    for _ in range(4):
        orientation = np.where(orientation >= tau, orientation - tau, orientation).astype(F)
    b = (orientation * bins_per_radian).astype(F)                    # :213
    exponent_numerator = rx * rx + ry * ry                           # :216-218
    wgt = np.exp(-exponent_numerator / exponent_denominator).astype(F)
    value = (magnitude * wgt).astype(F)

    # addFeature, :82-117: the eight addValue calls of a sample, in the source's order
    fx, cx, fy, cy = np.floor(bx), np.ceil(bx), np.floor(by), np.ceil(by)
    ba, bb = np.floor(b).astype(np.int64), np.ceil(b).astype(np.int64)
    i_max = (bx - fx).astype(F); i_min = (F(1) - i_max).astype(F)
    j_max = (by - fy).astype(F); j_min = (F(1) - j_max).astype(F)
    b_max = (b - np.floor(b)).astype(F); b_min = (F(1) - b_max).astype(F)
    calls = [(fx, fy, ba, (i_min * j_min * b_min) * value), (fx, fy, bb, (i_min * j_min * b_max) * value),      # ca
             (cx, fy, ba, (i_max * j_min * b_min) * value), (cx, fy, bb, (i_max * j_min * b_max) * value),      # cb
             (cx, cy, ba, (i_max * j_max * b_min) * value), (cx, cy, bb, (i_max * j_max * b_max) * value),      # cc
             (fx, cy, ba, (i_min * j_max * b_min) * value), (fx, cy, bb, (i_min * j_max * b_max) * value)]      # cd
    X = np.stack([c[0] for c in calls], axis=1).astype(np.int64).ravel()      # sample-major, the 8 calls of a sample together
    Y = np.stack([c[1] for c in calls], axis=1).astype(np.int64).ravel()
    B = np.stack([c[2] for c in calls], axis=1).ravel()
    V = np.stack([c[3] for c in calls], axis=1).astype(F).ravel()
    keep = (X >= 0) & (X < d) & (Y >= 0) & (Y < d)                   # addValue, :68-70
    B = np.where(B < 0, B + bins, B)                                 # :71-76 (each test once)
    B = np.where(B >= bins, B - bins, B)
    features = np.zeros(d * d * bins, F)
    np.add.at(features, ((Y * d * bins) + (X * bins) + B)[keep], V[keep])      # offset(), :53-57

    f = _normalize(features)                                         # :227-230
    f = np.minimum(f, F(0.2))
    f = _normalize(f)
    q = np.minimum(F(255.0), f * F(512.0)).astype(np.int32)          # quantizeFeatures :42-50: float -> int truncates
    return q, f
