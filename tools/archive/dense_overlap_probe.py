"""Probe: does the dense-texture step overlap its HBM-bound pyramid with its VALU-bound keypoint stages better when the ring blur
leaves LDS free on every CU?  (experiment build: -DSIFTMI_EXPERIMENT, SIFTMI_EXP_RING_PAD_LDS=<bytes>)
usage: SIFTMI_LIB=tools/tmp_variants/libsiftmi_exp.so python tools/dense_overlap_probe.py [dense|sparse] [pipeline] [graph_fork]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import siftmetal_amd as sm
from siftmetal_amd import _capi, stream as smstream
import bench

kind = sys.argv[1] if len(sys.argv) > 1 else "dense"
pipeline = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fork = int(sys.argv[3]) if len(sys.argv) > 3 else 0
F, W, H = 64, 1920, 1080
frames = bench.make_dense_frames(F) if kind == "dense" else bench.make_frames(F, 8)
d = smstream.DeviceFrames(frames)
eng = sm.Engine(W, H, n_octaves=4, max_batch=F, graph_fork=fork)
fs = smstream.FrameStream(eng, F, pipeline=pipeline, result_sets=2 * pipeline)
for _ in range(3 * fs.n_sets):
    fs.run(d)
_capi.check(_capi.load().siftmi_device_synchronize(0))
res = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        fs.run(d)
    _capi.check(_capi.load().siftmi_device_synchronize(0))
    res.append((time.perf_counter() - t0) / 10 * 1e3)
r = fs.results_host()
print("%s pipeline %d fork %d pad %s: %.3f ms/step (best of 3: %s)  %d descriptors" %
      (kind, pipeline, fork, os.environ.get("SIFTMI_EXP_RING_PAD_LDS", "0"), min(res), ["%.3f" % x for x in res], r["n_descriptors"]))
