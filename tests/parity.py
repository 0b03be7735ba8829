"""Comparison helpers: HIP path (siftmetal_amd.Engine) vs the CPU oracle, with the tolerances of
SURVEY.md section 8c written out.  Used by tests/ and __graft_entry__.smoke()."""
import numpy as np

# --- tolerances (float-order sensitivity measured in its Appendix C) ---------------
TOL_PYRAMID = 0.0          # Gaussian stack: bit-exact (same tap order, fmaf on both sides)
TOL_ABS_PX = 5e-3          # |d absoluteCoordinate| px
TOL_SUBSCALE = 1e-3
TOL_VALUE = 1e-6
TOL_SIGMA_REL = 1e-5
TOL_THETA = 2e-3           # rad
MAX_DESC_BIN_DIFF = 1      # quantised features: |d| <= 1 ...
MAX_DESC_BIN_FRAC = 1e-3   # ... on <= 0.1 % of bins
TOL_DESC_L2 = 1e-4         # pre-quantisation unit-norm float vector


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
    same_count, max_dt, n_cmp, mism = 0, 0.0, 0, 0
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
            max_dt = max(max_dt, float(ang_diff(g_ori["orientations"][k][:c], th).max()))
            n_cmp += c
    return {"n_kp": n_kp, "count_mismatch": mism, "max_dtheta": max_dt, "angles_compared": n_cmp}


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
