"""Comparison helpers: HIP path (siftmetal_amd.Engine) vs the CPU oracle, with the tolerances of
SURVEY.md section 8c written out.  Used by tests/ and __graft_entry__.smoke()."""
import os

import numpy as np

# --- tolerances (float-order sensitivity measured in its Appendix C) ---------------
TOL_PYRAMID = 0.0          # Gaussian stack: bit-exact (same tap order, fmaf on both sides)
TOL_ABS_PX = 5e-3          # |d absoluteCoordinate| px
TOL_SUBSCALE = 1e-3
TOL_VALUE = 1e-6
TOL_SIGMA_REL = 1e-5
TOL_THETA = 2e-3           # rad
MAX_DESC_BIN_DIFF = 1      # quantised features: |d| <= 1 ...
MAX_DESC_BIN_FRAC = 1e-3   # ... on <= 0.1 % of bins (bins_allowed: rounded up to a whole bin, so that a case with a handful of
                           # descriptors may hold ONE knife-edge bin -- a float within 1e-4 L2 straddling an integer -- as well)
TOL_DESC_L2 = 1e-4         # pre-quantisation unit-norm float vector


def bins_allowed(n_bins):
    """Quantised bins that may differ (by 1) among n_bins compared: 0.1 % -- rounded UP to one whole bin only for comparisons of fewer than
    1000 bins (a handful of descriptors may hold ONE knife-edge bin; round 4's rule rounded up at every size), down otherwise."""
    return 1 if 0 < n_bins < 1000 else int(MAX_DESC_BIN_FRAC * n_bins)


def prefilter_extrema(orc, o, ext, dog_threshold=0.0133, border=5):
    """The candidates the HIP extrema kernel emits = raw extrema that survive the refinement-entry
    tests of the reference (SIFTInterpolate.metal:208, :223)."""
    if len(ext) == 0:
        return ext
    w, h = orc.octave_size(o)
    v = np.array([orc.dog(o, int(s))[int(y), int(x)] for x, y, s in zip(ext["x"], ext["y"], ext["scale"])], np.float32)
    keep = np.abs(v) > np.float32(dog_threshold) * np.float32(0.8)
    keep &= (ext["x"] >= border) & (ext["x"] <= w - border - 1) & (ext["y"] >= border) & (ext["y"] <= h - border - 1)
    return ext[keep]


def ext_set(e):
    return set(zip(e["scale"].tolist(), e["y"].tolist(), e["x"].tolist()))


def kp_key(k, gpu):
    return (k["scale"], k["y"], k["x"])


def match_keypoints(g, r):
    """g: GPU keypoints (siftmi dtype), r: oracle keypoints (oracle dtype) of ONE octave.
    Returns (pairs [(gi, ri)], n_only_gpu, n_only_ref).  Matching is on the integer (scale, y, x)
    position plus nearest absolute coordinate (duplicates exist: reference keeps them)."""
    from collections import defaultdict
    buckets = defaultdict(list)
    for i in range(len(r)):
        buckets[(int(r["scale"][i]), int(r["y"][i]), int(r["x"][i]))].append(i)
    pairs, only_g = [], 0
    for i in range(len(g)):
        key = (int(g["scale"][i]), int(g["y"][i]), int(g["x"][i]))
        cand = buckets.get(key)
        if not cand:
            only_g += 1
            continue
        d = [abs(float(g["abs_x"][i]) - float(r["absX"][j])) + abs(float(g["abs_y"][i]) - float(r["absY"][j])) for j in cand]
        j = cand.pop(int(np.argmin(d)))
        pairs.append((i, j))
    only_r = sum(len(v) for v in buckets.values())
    return pairs, only_g, only_r


def compare_keypoints(g, r):
    pairs, og, orr = match_keypoints(g, r)
    n = max(len(g), len(r), 1)
    rep = {"n_gpu": len(g), "n_ref": len(r), "matched": len(pairs), "only_gpu": og, "only_ref": orr,
           "set_agreement": len(pairs) / n}
    if pairs:
        gi = np.array([p[0] for p in pairs]); ri = np.array([p[1] for p in pairs])
        rep["max_abs_px"] = float(max(np.abs(g["abs_x"][gi] - r["absX"][ri]).max(), np.abs(g["abs_y"][gi] - r["absY"][ri]).max()))
        rep["max_subscale"] = float(np.abs(g["sub_scale"][gi] - r["subScale"][ri]).max())
        rep["max_value"] = float(np.abs(g["value"][gi] - r["value"][ri]).max())
        rep["max_sigma_rel"] = float((np.abs(g["sigma"][gi] - r["sigma"][ri]) / r["sigma"][ri]).max())
        rep["max_norm"] = float(max(np.abs(g["norm_x"][gi] - r["normX"][ri]).max(), np.abs(g["norm_y"][gi] - r["normY"][ri]).max()))
    return rep, pairs


def ang_diff(a, b):
    return np.abs((a - b + np.pi) % (2 * np.pi) - np.pi)


def compare_orientations(g_ori, r_ori, n_kp):
    """g_ori: siftmi_orientation records for all keypoints of the octave (count -1 = rejected);
    r_ori: oracle records (only the keypoints that passed the border filter)."""
    rc = {int(k): (int(c), th[:c]) for k, c, th in zip(r_ori["keypoint"], r_ori["count"], r_ori["orientations"])}
    same_count, max_dt, n_cmp, mism, n_over = 0, 0.0, 0, 0, 0
    for k in range(n_kp):
        gc = int(g_ori["count"][k])
        if k not in rc:
            if gc != -1:
                mism += 1
            continue
        c, th = rc[k]
        if gc != c:
            mism += 1
            continue
        same_count += 1
        if c:
            d = ang_diff(g_ori["orientations"][k][:c], th)
            max_dt = max(max_dt, float(d.max()))
            n_over += int((d > TOL_THETA).sum())
            n_cmp += c
    return {"n_kp": n_kp, "count_mismatch": mism, "max_dtheta": max_dt, "angles_compared": n_cmp, "over_tol": n_over}


def compare_descriptors(g_desc, g_f32, r_desc, r_f32, r_ori):
    """Descriptors of ONE octave computed from IDENTICAL keypoints.  GPU descriptor.keypoint indexes
    the keypoint list; the oracle's indexes its orientation list -> map through r_ori.keypoint.
    Pairs by (keypoint, nearest theta)."""
    from collections import defaultdict
    rb = defaultdict(list)
    for i in range(len(r_desc)):
        rb[int(r_ori["keypoint"][int(r_desc["keypoint"][i])])].append(i)
    n_bins = n_diff = 0
    max_bin = 0
    max_l2 = 0.0
    unmatched = 0
    max_dt = 0.0
    for i in range(len(g_desc)):
        cand = rb.get(int(g_desc["keypoint"][i]))
        if not cand:
            unmatched += 1
            continue
        dt = ang_diff(np.float64(g_desc["theta"][i]), r_desc["theta"][cand].astype(np.float64))
        jj = int(np.argmin(dt))
        if dt[jj] > 0.05:
            unmatched += 1
            continue
        j = cand.pop(jj)
        max_dt = max(max_dt, float(dt[jj]))
        d = np.abs(g_desc["features"][i].astype(np.int32) - r_desc["features"][j].astype(np.int32))
        n_bins += 128
        n_diff += int((d > 0).sum())
        max_bin = max(max_bin, int(d.max()))
        if g_f32 is not None and r_f32 is not None:
            max_l2 = max(max_l2, float(np.sqrt(((g_f32[i].astype(np.float64) - r_f32[j].astype(np.float64)) ** 2).sum())))
    unmatched += sum(len(v) for v in rb.values())
    return {"n_gpu": len(g_desc), "n_ref": len(r_desc), "unmatched": unmatched, "bins": n_bins, "bins_differing": n_diff,
            "frac_differing": n_diff / max(n_bins, 1), "max_bin_diff": max_bin, "max_l2_float": max_l2, "max_dtheta": max_dt}


def to_oracle_keypoints(g):
    """siftmi keypoint records -> oracle keypoint records (same fields, other names)."""
    from oracle import pyoracle
    out = np.zeros(len(g), pyoracle.keypoint_dtype)
    for a, b in [("octave", "octave"), ("scale", "scale"), ("subScale", "sub_scale"), ("x", "x"), ("y", "y"), ("absX", "abs_x"),
                 ("absY", "abs_y"), ("normX", "norm_x"), ("normY", "norm_y"), ("sigma", "sigma"), ("value", "value")]:
        out[a] = g[b]
    return out


def to_oracle_orientations(g_ori):
    """siftmi_orientation records (count -1 = rejected by the border filter) -> the oracle's list of
    accepted keypoints, so that the descriptor stage can be compared on bit-identical (keypoint, theta)."""
    from oracle import pyoracle
    keep = g_ori["count"] >= 0
    out = np.zeros(int(keep.sum()), pyoracle.orientation_dtype)
    out["keypoint"] = g_ori["keypoint"][keep]
    out["count"] = g_ori["count"][keep]
    out["orientations"] = g_ori["orientations"][keep]
    return out


def _split(arr, counts):
    out, pos = [], 0
    for c in counts:
        out.append(arr[pos:pos + c])
        pos += c
    return out


def check_full_path(sm, img, no, nspo, strict_theta=True, expect=None, symmetric_pattern=False, **engine_kw):
    """Every stage of the HIP path against the oracle on one image (used by tests/test_gpu_parity.py and
    tools/fuzz_parity.py).  Raises AssertionError with the failing stage.

    strict_theta=False (the randomised sweep): the reference's orientation histogram assigns each sample to the NEAREST bin
    (round(36 theta / 2 pi), SIFTOrientation.metal:122-129), so it is discontinuous in the gradient angle: where a sample's
    angle sits within an ulp of a bin boundary, two correct atan2f implementations put it into different bins, the histogram
    changes by one whole sample (~1e-3 of a peak) and an interpolated peak on a flat histogram moves by up to ~1e-2 rad.
    That happens to a fraction of a percent of the angles; the sweep therefore allows 2 % of the angles to exceed TOL_THETA,
    none by more than 0.05 rad (under a third of a bin), instead of a hard maximum.

    symmetric_pattern=True (checkerboards in the sweep): an exactly symmetric image puts 15 % of its gradients within 1e-6 of |dx| == |dy|, i.e.
    ON the 45 / 135 degree boundaries of the 36-bin histogram, and gives every corner four equal peaks; which side such a sample falls on is
    decided in the last ulp of atan2f in the reference's own f32 expression, every flipped sample is ~1 % of a peak, and the interpolated peak
    moves by up to ~1e-2 rad on about half of such a case's angles (rounds 5 and 6 alike: profiles/sweep_cases_r06.log).  Only the hard limit
    of the sweep (0.05 rad, under a third of a bin) is kept for the angles of such a case.  Peak COUNTS are reported, not asserted, there: a peak
    is a strict local maximum of the smoothed histogram (SIFTOrientation.metal:150-168), and where the symmetry makes two adjacent bins equal to
    the last bit, whether there is a peak at all depends on that bit (sweep seed 424242, case 67, a 449 x 323 checkerboard: 4196 of 8769 keypoints
    differ in their number of orientations from the oracle -- with the round-5 library exactly as with this one: profiles/sweep_cases_r06.log).
    Every other stage is checked as usual, the descriptors from the GPU's own (keypoint, theta) list.

    expect: for a FIXED case, the mismatch counts observed on it -- {"orientation_count_mismatch": n, "descriptors_unmatched": m,
    "bins_differing": b} summed over the octaves -- asserted exactly (SURVEY 8c grants "same count", not a budget);
    None (the randomised sweep, new cases) keeps the budget of one per 200 keypoints.  The observed counts are returned."""
    import sys
    from oracle import pyoracle
    parity = sys.modules[__name__]
    h, w = img.shape[:2]
    eng = sm.Engine(w, h, n_octaves=no, nspo=nspo, keep_descriptor_floats=1, **engine_kw)
    orc = pyoracle.Oracle(w, h, n_octaves=no, nspo=nspo)
    NG = nspo + 3
    ref = orc.run(img, want_float=True)

    kps, kc, ds, dc = eng.detect_describe_batch(img[None])
    st = eng.stats()
    g_kp, g_ds = _split(kps, kc[0]), _split(ds, dc[0])

    # schedule
    for o in range(no):
        assert eng.octave_size(o)[:2] == orc.octave_size(o) and eng.octave_size(o)[2] == orc.delta(o)
        for s in range(NG):
            assert eng.sigma(o, s) == orc.sigma(o, s)
    for l in range(NG):
        assert np.array_equal(eng.weights(l), orc.weights(l))

    tot_kp = tot_match = 0
    seen = {"orientation_count_mismatch": 0, "descriptors_unmatched": 0, "bins_differing": 0, "max_dtheta": 0.0, "max_l2_float": 0.0}
    for o in range(no):
        # 1. Gaussian stack: bit-exact
        for s in range(NG):
            G, R = eng.gaussian(o, s), orc.gaussian(o, s)
            assert np.array_equal(G, R), "octave %d layer %d: max |d| = %g" % (o, s, np.abs(G - R).max())
        # 2. extrema: identical raw count, identical candidate set
        # raw count: exact unless the extrema scan skips rows flagged inactive by the marching blur (cfg.count_raw_extrema = 0
        # on launches that use it; small images only do when blur_march_min_blocks forces it)
        # (... or, round 3, a single frame with octaves of >= 1.5 Mpixel, whose tile blur writes the flags too); the library says which
        if engine_kw.get("count_raw_extrema", 0) or ("blur_march_min_blocks" not in engine_kw and img.shape[0] * img.shape[1] * 4 < 1500000):
            assert st["raw_extrema_exact"]
        if st["raw_extrema_exact"]:
            assert st["raw_extrema"][0, o] == len(ref[o]["extrema"])
        else:
            assert st["raw_extrema"][0, o] <= len(ref[o]["extrema"])
        cand = parity.prefilter_extrema(orc, o, ref[o]["extrema"])
        assert parity.ext_set(eng.extrema(o)) == parity.ext_set(cand)
        assert st["candidates"][0, o] == len(cand)
        # 3. keypoints
        rep, pairs = parity.compare_keypoints(g_kp[o], ref[o]["keypoints"])
        tot_kp += max(rep["n_gpu"], rep["n_ref"]); tot_match += rep["matched"]
        if pairs:
            assert rep["max_abs_px"] <= parity.TOL_ABS_PX and rep["max_subscale"] <= parity.TOL_SUBSCALE
            assert rep["max_value"] <= parity.TOL_VALUE and rep["max_sigma_rel"] <= parity.TOL_SIGMA_REL
            assert rep["max_norm"] == 0.0
        assert (g_kp[o]["octave"] == o).all()
        # sorted by (scale, y, x)
        key = g_kp[o]["scale"].astype(np.int64) * (1 << 40) + g_kp[o]["y"].astype(np.int64) * (1 << 20) + g_kp[o]["x"]
        assert (np.diff(key) >= 0).all()
        # 4./5. orientation + descriptors from IDENTICAL keypoints (the GPU's), so that float noise
        # in refinement does not leak into the comparison of these stages
        okp = parity.to_oracle_keypoints(g_kp[o])
        r_ori = orc.orientations(o, okp)
        g_ori = eng.orientations(o)
        orep = parity.compare_orientations(g_ori, r_ori, len(okp))
        if expect is None and not symmetric_pattern:
            assert orep["count_mismatch"] <= max(1, len(okp) // 200), orep
        seen["orientation_count_mismatch"] += orep["count_mismatch"]
        seen["max_dtheta"] = max(seen["max_dtheta"], orep["max_dtheta"])
        seen["angles_over_tol"] = seen.get("angles_over_tol", 0) + orep["over_tol"]
        seen["angles_compared"] = seen.get("angles_compared", 0) + orep["angles_compared"]
        if strict_theta:
            assert orep["max_dtheta"] <= parity.TOL_THETA, orep
        elif symmetric_pattern:
            assert orep["max_dtheta"] <= 0.05, orep
        else:
            assert orep["over_tol"] <= max(1, orep["angles_compared"] // 50) and orep["max_dtheta"] <= 0.05, orep
        assert st["oriented"][0, o] == int((g_ori["count"] >= 0).sum())
        # ... and the descriptor stage from the GPU's own (keypoint, theta) list, bit-identical inputs
        in_ori = parity.to_oracle_orientations(g_ori)
        r_desc, r_f32 = orc.descriptors(o, okp, in_ori, want_float=True)
        drep = parity.compare_descriptors(g_ds[o], eng.descriptor_floats(o), r_desc, r_f32, in_ori)
        assert drep["max_dtheta"] == 0.0 and drep["n_gpu"] == drep["n_ref"], drep
        if expect is None:
            assert drep["unmatched"] <= 2 * max(1, len(okp) // 200), drep
        seen["descriptors_unmatched"] += drep["unmatched"]
        seen["bins_differing"] += drep["bins_differing"]
        seen["max_l2_float"] = max(seen["max_l2_float"], drep["max_l2_float"])
        assert drep["max_bin_diff"] <= parity.MAX_DESC_BIN_DIFF, drep
        assert drep["bins_differing"] <= parity.bins_allowed(drep["bins"]), drep
        assert drep["max_l2_float"] <= parity.TOL_DESC_L2, drep
    assert tot_match >= 0.995 * tot_kp - 1, (tot_match, tot_kp)
    eng.close()
    seen.update({"keypoints": int(tot_kp), "matched": int(tot_match)})
    if os.environ.get("SIFTMI_PARITY_LOG"):
        with open(os.environ["SIFTMI_PARITY_LOG"], "a") as f:
            f.write("%s %dx%d no=%d nspo=%d %s %s\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], w, h, no, nspo, sorted(engine_kw.items()), seen))
    if expect is not None:
        for k, v in expect.items():
            assert seen[k] == v, (k, seen[k], v, seen)
    return seen
