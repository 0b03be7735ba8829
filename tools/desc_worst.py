"""The descriptors of a sweep case that differ most from the oracle's: unit-vector L2, integers off, where the window lies.
usage: [SIFTMI_LIB=<build>] python tools/desc_worst.py <seed> <n_cases_generated> <index> [top]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import siftmetal_amd as sm
from oracle import pyoracle
from tests import parity, sweep

seed, n, idx = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 4
c = sweep.parity_cases(seed, n, nspo_choices=(3, 3, 4, 5, 6, 7))[idx]
img = c["img"]; h, w = img.shape[:2]
eng = sm.Engine(w, h, n_octaves=c["octaves"], nspo=c["nspo"], keep_descriptor_floats=1, max_extrema=1 << 18, max_keypoints=1 << 17, max_descriptors=1 << 19, **dict(c["mode"]))
orc = pyoracle.Oracle(w, h, n_octaves=c["octaves"], nspo=c["nspo"])
orc.run(img, want_float=True)
kps, kc, ds, dc = eng.detect_describe_batch(img[None])
print(os.path.basename(os.environ.get("SIFTMI_LIB", "libsiftmi.so")), "case", idx, sweep.describe_case(c))
pos = dpos = 0
for o in range(c["octaves"]):
    g = kps[pos:pos + kc[0, o]]
    okp = parity.to_oracle_keypoints(g)
    in_ori = parity.to_oracle_orientations(eng.orientations(o))
    r_desc, r_f32 = orc.descriptors(o, okp, in_ori, want_float=True)
    gd, gf = ds[dpos:dpos + dc[0, o]], eng.descriptor_floats(o)
    ow, oh, delta = eng.octave_size(o)
    rows = []
    for i in range(min(len(gd), len(r_desc))):          # same (keypoint, theta) order on both sides when the counts agree
        if int(in_ori["keypoint"][int(r_desc["keypoint"][i])]) != int(gd["keypoint"][i]):
            continue
        l2 = float(np.sqrt(((gf[i].astype(np.float64) - r_f32[i].astype(np.float64)) ** 2).sum()))
        nd = int((gd["features"][i].astype(np.int32) != r_desc["features"][i].astype(np.int32)).sum())
        k = g[int(gd["keypoint"][i])]
        hw = 3.0 * 1.6 * 2.0 ** ((float(k["scale"]) + float(k["sub_scale"])) / c["nspo"])
        radius = int(hw * 2 ** 0.5 * 2.5 + 0.5)
        px, py = int(k["abs_x"]) / delta, int(k["abs_y"]) / delta
        interior = px - radius >= 1 and px + radius <= ow - 2 and py - radius >= 1 and py + radius <= oh - 2
        rows.append((l2, nd, o, i, round(px, 1), round(py, 1), radius, interior, float(k["value"])))
    rows.sort(reverse=True)
    print("  octave %d: %d descriptors, sum of float vectors %.9f, mean L2 %.4g" % (o, len(gd), float(gf[:len(gd)].astype(np.float64).sum()), float(np.mean([r[0] for r in rows])) if rows else 0.0))
    if rows and os.environ.get("DESC_WORST_DUMP"):
        i = rows[0][3]
        d = (gf[i].astype(np.float64) - r_f32[i].astype(np.float64)).reshape(16, 8)
        print("  worst descriptor's difference x 1e6 by cell (rows) and bin (columns); oracle vector max %.4f min %.5f" % (r_f32[i].max(), r_f32[i].min()))
        for r in range(16):
            print("   ", " ".join("%7.2f" % (v * 1e6) for v in d[r]), "   |", " ".join("%.4f" % v for v in r_f32[i].reshape(16, 8)[r]))
    for r in rows[:top]:
        print("  octave %d (%dx%d): L2 %.3g, %d ints off, descriptor %d at (%.1f, %.1f) radius %d interior %s DoG value %.4f" % (r[2], ow, oh, r[0], r[1], r[3], r[4], r[5], r[6], r[7], r[8]))
    pos += kc[0, o]; dpos += dc[0, o]
eng.close()
