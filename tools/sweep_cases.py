"""Selected cases of the parity sweep (tests/sweep.py), every stage against the oracle, with the orientation / descriptor agreement printed even
where a case fails -- for comparing builds on the cases that are hard (exactly symmetric patterns, low-contrast descriptors).
usage: [SIFTMI_LIB=<build>] python tools/sweep_cases.py <seed> <n_cases_generated> <index> [<index> ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import siftmetal_amd as sm
from oracle import pyoracle
from tests import parity, sweep

seed, n = int(sys.argv[1]), int(sys.argv[2])
cases = sweep.parity_cases(seed, n, nspo_choices=(3, 3, 4, 5, 6, 7))
for idx in map(int, sys.argv[3:]):
    c = cases[idx]
    img = c["img"]; h, w = img.shape[:2]
    kw = dict(c["mode"])
    eng = sm.Engine(w, h, n_octaves=c["octaves"], nspo=c["nspo"], keep_descriptor_floats=1, max_extrema=1 << 18, max_keypoints=1 << 17, max_descriptors=1 << 19, **kw)
    orc = pyoracle.Oracle(w, h, n_octaves=c["octaves"], nspo=c["nspo"])
    orc.run(img, want_float=True)
    kps, kc, ds, dc = eng.detect_describe_batch(img[None])
    pos = dpos = 0
    tot = {"n_kp": 0, "count_mismatch": 0, "angles": 0, "over_tol": 0, "max_dtheta": 0.0, "bins": 0, "bins_differing": 0, "max_l2": 0.0}
    for o in range(c["octaves"]):
        g = kps[pos:pos + kc[0, o]]
        okp = parity.to_oracle_keypoints(g)
        g_ori = eng.orientations(o)
        orep = parity.compare_orientations(g_ori, orc.orientations(o, okp), len(okp))
        in_ori = parity.to_oracle_orientations(g_ori)
        r_desc, r_f32 = orc.descriptors(o, okp, in_ori, want_float=True)
        drep = parity.compare_descriptors(ds[dpos:dpos + dc[0, o]], eng.descriptor_floats(o), r_desc, r_f32, in_ori)
        tot["n_kp"] += len(okp); tot["count_mismatch"] += orep["count_mismatch"]; tot["angles"] += orep["angles_compared"]; tot["over_tol"] += orep["over_tol"]
        tot["max_dtheta"] = max(tot["max_dtheta"], orep["max_dtheta"]); tot["bins"] += drep["bins"]; tot["bins_differing"] += drep["bins_differing"]
        tot["max_l2"] = max(tot["max_l2"], drep["max_l2_float"])
        pos += kc[0, o]; dpos += dc[0, o]
    print("%s case %d %s -> %s" % (os.path.basename(os.environ.get("SIFTMI_LIB", "libsiftmi.so")), idx, sweep.describe_case(c),
                                   {k: (float("%.3g" % v) if isinstance(v, float) else v) for k, v in tot.items()}), flush=True)
    eng.close()
