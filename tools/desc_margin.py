"""Descriptor / orientation parity margins of the HIP path against the oracle (run on the GPU box): how far inside the
stated tolerances (tests/parity.py) the float descriptor, its quantised integers and theta sit.
usage: python tools/desc_margin.py [dense]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

import siftmetal_amd as sm
from oracle import pyoracle
from tests import parity
from tests.synth import blob_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
im = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "butterfly.png")))
b = np.ascontiguousarray(im[..., [2, 1, 0, 3]])
cases = [("butterfly", b, 5), ("blob1080p", blob_frame(1920, 1080, 0), 4)]
if len(sys.argv) > 1 and sys.argv[1] == "dense":
    row = np.concatenate([b, b[:, ::-1], b, b[:, ::-1]], axis=1)
    cases.append(("dense1080p", np.ascontiguousarray(np.concatenate([row, row[::-1], row, row[::-1]], axis=0)[:1080, :1920]), 4))
for name, img, no in cases:
    h, w = img.shape[:2]
    eng = sm.Engine(w, h, n_octaves=no, keep_descriptor_floats=1)
    orc = pyoracle.Oracle(w, h, n_octaves=no)
    orc.build_pyramid(img)
    kps, kc, ds, dc = eng.detect_describe_batch(img[None])
    pos = dpos = 0
    worst = {"max_l2_float": 0.0, "frac_differing": 0.0, "max_bin_diff": 0, "max_dtheta_ori": 0.0, "n": 0, "bins_differing": 0, "bins": 0}
    t0 = time.time()
    for o in range(no):
        g = kps[pos:pos + kc[0, o]]
        okp = parity.to_oracle_keypoints(g)
        g_ori = eng.orientations(o)
        orep = parity.compare_orientations(g_ori, orc.orientations(o, okp), len(okp))
        in_ori = parity.to_oracle_orientations(g_ori)
        r_desc, r_f32 = orc.descriptors(o, okp, in_ori, want_float=True)
        drep = parity.compare_descriptors(ds[dpos:dpos + dc[0, o]], eng.descriptor_floats(o), r_desc, r_f32, in_ori)
        worst["max_l2_float"] = max(worst["max_l2_float"], drep["max_l2_float"])
        worst["max_bin_diff"] = max(worst["max_bin_diff"], drep["max_bin_diff"])
        worst["bins_differing"] += drep["bins_differing"]; worst["bins"] += drep["bins"]
        worst["max_dtheta_ori"] = max(worst["max_dtheta_ori"], orep["max_dtheta"])
        worst["n"] += drep["n_gpu"]
        pos += kc[0, o]; dpos += dc[0, o]
    worst["frac_differing"] = worst["bins_differing"] / max(worst["bins"], 1)
    print(name, {k: (float("%.3g" % v) if isinstance(v, float) else v) for k, v in worst.items()}, "(tolerances: L2 1e-4, frac 1e-3, bin 1, theta 2e-3)", flush=True)
    eng.close()
