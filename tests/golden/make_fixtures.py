"""Builds tests/golden/ from the data files the reference's own tests hold.

Run ONLY in the development container (reads /root/reference, which does not exist on the GPU
box).  Everything written here is DATA -- the reference's test inputs and IPOL-produced expected
outputs (Tests/SIFTMetalTests/Resources) -- repacked into small arrays; no reference source text.

    python tests/golden/make_fixtures.py

Outputs
  butterfly.png                   verbatim copy of the test image (512x340 RGBA8)
  butterfly_ipol.npz
     nes / dog_soft / extr_interp / dog_thresh / on_edge   float32 [n,3] = (y, x, sigma) rows of
                                   extra_{NES,DoGSoftThresh,ExtrInterp,DoGThresh,OnEdgeResp}_butterfly.txt
     desc_yxst                     float32 [1609,4] = y x sigma theta of butterfly-descriptors.txt
     desc_features                 uint8   [1609,128]
     desc_orihist                  float32 [1609,36]
     scalespace_o{o}_s{s}          uint8 [h_o, w_o]: scalespace_butterfly_o00o_s00s.png de-upsampled
                                   by [::2^o, ::2^o] (the PNGs are all NN-upsampled to 1024x680)
"""
import os
import shutil

import numpy as np
from PIL import Image

SRC = "/root/reference/Tests/SIFTMetalTests/Resources"
DST = os.path.dirname(os.path.abspath(__file__))


def load_rows(name):
    rows = []
    with open(os.path.join(SRC, name)) as f:
        for line in f:
            p = line.split()
            if p:
                rows.append([float(v) for v in p])
    return np.array(rows, dtype=np.float64)


def main():
    shutil.copyfile(os.path.join(SRC, "butterfly.png"), os.path.join(DST, "butterfly.png"))
    out = {}
    for key, name in [("nes", "extra_NES_butterfly.txt"), ("dog_soft", "extra_DoGSoftThresh_butterfly.txt"),
                      ("extr_interp", "extra_ExtrInterp_butterfly.txt"), ("dog_thresh", "extra_DoGThresh_butterfly.txt"),
                      ("on_edge", "extra_OnEdgeResp_butterfly.txt"), ("far_from_border", "extra_FarFromBorder_butterfly.txt")]:
        out[key] = load_rows(name)[:, :3].astype(np.float32)
    d = load_rows("butterfly-descriptors.txt")
    assert d.shape == (1609, 4 + 128 + 36), d.shape
    out["desc_yxst"] = d[:, :4].astype(np.float32)
    out["desc_features"] = d[:, 4:132].astype(np.uint8)
    out["desc_orihist"] = d[:, 132:].astype(np.float32)
    for o in range(5):
        for s in range(6):
            im = np.array(Image.open(os.path.join(SRC, "scalespace_butterfly_o%03d_s%03d.png" % (o, s))))
            if im.ndim == 3:
                im = im[..., 0]
            assert im.shape == (680, 1024), im.shape
            st = 2 ** o
            out["scalespace_o%d_s%d" % (o, s)] = np.ascontiguousarray(im[::st, ::st]).astype(np.uint8)
    # wire-format sample: the first 8 lines of the descriptor file and of one keypoint file, verbatim (data, text form)
    for name, dst in [("butterfly-descriptors.txt", "butterfly-descriptors-head.txt"),
                      ("extra_OnEdgeResp_butterfly.txt", "butterfly-keypoints-head.txt")]:
        with open(os.path.join(SRC, name)) as f, open(os.path.join(DST, dst), "w") as g:
            for i, line in enumerate(f):
                if i == 8:
                    break
                g.write(line)
    np.savez_compressed(os.path.join(DST, "butterfly_ipol.npz"), **out)
    print("wrote", {k: v.shape for k, v in out.items() if not k.startswith("scalespace")})


if __name__ == "__main__":
    main()
