"""Rows after describe: derived descriptor vectors, the text wire format, compareGeometry.
CPU-only: the oracle against literal expectations / the reference's fixture files, and the library's host-side
functions (no device needed) against the oracle."""
import os

import numpy as np
import pytest

from oracle import pyoracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sift_like(rng, n, spread=40.0):
    return np.clip(np.abs(rng.normal(0.0, spread, (n, 128))), 0, 255).astype(np.int32)


# ---------------------------------------------------------------- SIFTDescriptor.init derived vectors

def test_index_vectors_oracle_layout():
    f = np.arange(128, dtype=np.int32)[None] * 2 % 256
    raw, val, key = pyoracle.descriptor_index(f)
    np.testing.assert_array_equal(raw[0], f[0].astype(np.float32) / np.float32(255))
    order = [5, 6, 9, 10, 0, 3, 12, 15, 1, 2, 4, 7, 8, 11, 13, 14]          # SIFTDescriptor.swift:49-73
    cells = raw[0].reshape(16, 8)
    np.testing.assert_array_equal(val[0].reshape(16, 8), cells[order])
    np.testing.assert_allclose(key[0], cells[order].mean(axis=1), rtol=3e-7)
    # a permutation of whole cells: distances are unchanged (what lets the matcher skip the re-ordering)
    g = sift_like(np.random.default_rng(1), 2)
    _, v, _ = pyoracle.descriptor_index(g)
    r = g.astype(np.float32) / np.float32(255)
    assert abs(np.sum((v[0] - v[1]) ** 2) - np.sum((r[0] - r[1]) ** 2)) < 1e-5


def test_index_vectors_library_equals_oracle():
    import siftmetal_amd as sm
    rng = np.random.default_rng(2)
    f = sift_like(rng, 50)
    raw, val, key = sm._index_vectors(f)
    oraw, oval, okey = pyoracle.descriptor_index(f)
    np.testing.assert_array_equal(raw, oraw)
    np.testing.assert_array_equal(val, oval)
    np.testing.assert_array_equal(key, okey)
    d = sm.SIFTDescriptor(keypoint=None, theta=0.5, features=f[3].tolist())
    np.testing.assert_array_equal(d.rawFeatures, oraw[3])
    np.testing.assert_array_equal(d.indexKey, okey[3])
    with pytest.raises(ValueError):
        sm.SIFTDescriptor(keypoint=None, theta=0.0, features=[])            # precondition(features.count > 0)


# ---------------------------------------------------------------- text wire format

def test_wire_format_parses_reference_files(ipol):
    from siftmetal_amd import wire
    text = open(os.path.join(GOLDEN, "butterfly-descriptors-head.txt")).read()
    ds = wire.parseDescriptors(text)
    hs = wire.parseOrientationHistograms(text)
    assert len(ds) == 8 and len(hs) == 8
    for i, d in enumerate(ds):
        y, x, s, t = ipol["desc_yxst"][i]
        assert d.keypoint.absoluteCoordinate == pytest.approx((x, y), rel=1e-6)
        assert d.keypoint.sigma == pytest.approx(s, rel=1e-6) and d.theta == pytest.approx(t, rel=1e-6)
        assert (d.keypoint.octave, d.keypoint.scale, d.keypoint.subScale, d.keypoint.scaledCoordinate, d.keypoint.value) == (0, 0, 0.0, (0, 0), 0.0)
        np.testing.assert_array_equal(d.features, ipol["desc_features"][i])
        np.testing.assert_allclose(hs[i], ipol["desc_orihist"][i], atol=1e-6)
    assert wire.formatDescriptors(ds, hs) == text                            # byte-exact round trip
    ktext = open(os.path.join(GOLDEN, "butterfly-keypoints-head.txt")).read()
    ks = wire.parseKeypoints(ktext)
    assert len(ks) == 8
    for i, k in enumerate(ks):
        y, x, s = ipol["on_edge"][i]
        assert k.absoluteCoordinate == pytest.approx((x, y), rel=1e-6) and k.sigma == pytest.approx(s, rel=1e-6)
    assert [l.split()[:3] for l in wire.formatKeypoints(ks).splitlines()] == [l.split()[:3] for l in ktext.splitlines()]
    assert wire.parseDescriptors("\n\n") == []
    with pytest.raises(ValueError):
        wire.parseDescriptors("1 2 3 4 5 6\n")


# ---------------------------------------------------------------- compareGeometry

def literal_compare_geometry(matches, sxy, txy, minimum_sample_size=7):
    """SIFTDescriptor.swift:162-296 written out with numpy float32 scalars."""
    f = np.float32
    clamp = lambda v: min(max(v, f(0)), f(1))                               # noqa: E731
    length = lambda v: np.sqrt(f(v[0] * v[0]) + f(v[1] * v[1]), dtype=f)    # noqa: E731
    hdot = lambda a, b: clamp(f(f(f(a[0] * b[0]) + f(a[1] * b[1])) * f(0.5)) + f(0.5))   # noqa: E731
    scores, total = [], f(0)
    for i in range(0, len(matches) - 3):
        m0, m1, m2, m3 = matches[i:i + 4]
        sb = sxy[m1["source"]] - sxy[m0["source"]]
        tb = txy[m1["target"]] - txy[m0["target"]]
        sbl, tbl = length(sb), length(tb)
        if not sbl >= 2 or not tbl >= 2:
            continue
        st = sxy[m3["source"]] - sxy[m2["source"]]
        tt = txy[m3["target"]] - txy[m2["target"]]
        stl, ttl = length(st), length(tt)
        if not stl >= 2 or not ttl >= 2:
            continue
        sr, tr = f(stl / sbl), f(ttl / tbl)
        sd = hdot(st * f(f(1) / stl), sb * f(f(1) / sbl))
        td = hdot(tt * f(f(1) / ttl), tb * f(f(1) / tbl))
        osim = f(f(1) - abs(f(sd - td)))
        ssim = clamp(f(sr / tr)) if sr < tr else clamp(f(tr / sr))
        sim = f(osim * ssim)
        scores.append(f(sim * sim))
        total = f(total + scores[-1])
    if len(scores) < minimum_sample_size:
        return 0.0
    mean = f(total / f(len(scores)))
    err = f(0)
    for s in scores:
        err = f(err + f(f(s - mean) * f(s - mean)))
    sd = np.sqrt(f(err / f(len(scores) - 1)), dtype=f)
    fs, fn = f(0), f(0)
    with np.errstate(invalid="ignore", divide="ignore"):
        for s in scores:
            if abs(f(f(s - mean) / sd)) <= 2:
                fs, fn = f(fs + s), f(fn + f(1))
        return float(f(fs / fn))


def _matches(n, rng, n_pts):
    m = np.zeros(n, pyoracle.match_dtype)
    m["source"] = rng.permutation(n_pts)[:n]
    m["target"] = m["source"]
    return m


def test_compare_geometry_oracle_vs_literal_and_invariances():
    rng = np.random.default_rng(4)
    n_pts = 120
    sxy = rng.uniform(0, 600, (n_pts, 2)).astype(np.float32)
    m = _matches(60, rng, n_pts)
    # same geometry -> every score is 1 -> sd = 0 -> z = NaN -> fairMean = 0/0 (reference behaviour: NaN)
    assert np.isnan(pyoracle.compare_geometry(m, sxy, sxy))
    # similarity transform (rotation + scale + shift) with a little noise: near 1
    th, sc = 0.7, 1.8
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]], np.float32) * sc
    txy = (sxy @ R.T + np.float32([30, -12]) + rng.normal(0, 0.5, sxy.shape)).astype(np.float32)
    s_sim = pyoracle.compare_geometry(m, sxy, txy)
    assert 0.9 < s_sim <= 1.0
    assert s_sim == pytest.approx(literal_compare_geometry(m, sxy, txy), rel=2e-6)
    # unrelated geometry: clearly lower
    rxy = rng.uniform(0, 600, (n_pts, 2)).astype(np.float32)
    s_rand = pyoracle.compare_geometry(m, sxy, rxy)
    assert s_rand == pytest.approx(literal_compare_geometry(m, sxy, rxy), rel=2e-6)
    assert s_rand < 0.6 < s_sim
    # short segments are skipped (minimumLength 2), too few usable samples -> 0
    close = (sxy * np.float32(0.001)).astype(np.float32)
    assert pyoracle.compare_geometry(m, close, close) == 0.0
    assert pyoracle.compare_geometry(m[:9], sxy, txy) == 0.0                 # 9 matches -> 6 windows < 7
    assert pyoracle.compare_geometry(m[:10], sxy, txy) > 0.0


def test_match_geometry_oracle_flow():
    rng = np.random.default_rng(6)
    tgt = sift_like(rng, 150)
    src = np.clip(tgt[:100] + rng.integers(-5, 6, (100, 128)), 0, 255).astype(np.int32)
    sxy = rng.uniform(0, 500, (100, 2)).astype(np.float32)
    txy = np.concatenate([sxy * np.float32(1.5) + rng.normal(0, 0.3, sxy.shape).astype(np.float32),
                          rng.uniform(0, 500, (50, 2)).astype(np.float32)])
    score, n = pyoracle.match_geometry(src, sxy, tgt, txy)
    m = pyoracle.match(src, tgt)
    assert n == len(m) >= 80
    assert score == pyoracle.compare_geometry(m[:80], sxy, txy) and score > 0.9     # first 80 matches only (:136)
    score, n = pyoracle.match_geometry(src[:5], sxy[:5], tgt, txy)                   # fewer than 7 matches -> 0 (:127-130)
    assert n <= 5 and score == 0.0
