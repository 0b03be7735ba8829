"""Forked against serial launch graph (siftmi_config.graph_fork = 1 / -1) as the descriptor density of the frames grows: 64 x 1920x1080
frames whose left part (a fraction p of the width) is the mirror-tiled natural texture of SURVEY.md 8d and whose rest is the
benchmark's blob field.  Two steps in flight, ms per step.  Run on the GPU box:  python tools/fork_density_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import siftmetal_amd as sm
from siftmetal_amd import _capi, stream as smstream
from tests.synth import blob_frame
import bench
W, H, F = 1920, 1080, 64
blob = np.stack([blob_frame(W, H, i) for i in range(8)])
dense = bench.make_dense_frames(8)
for p in (0.0, 0.15, 0.3, 0.5, 0.75, 1.0):
    cut = int(W * p) // 4 * 4
    fr = blob.copy()
    fr[:, :, :cut] = dense[:, :, :cut]
    d = smstream.DeviceFrames(np.concatenate([fr] * 8))
    line = []
    nd = 0
    for rep in range(2):
        for mode in (1, -1):
            eng = sm.Engine(W, H, n_octaves=4, max_batch=F, graph_fork=mode)
            fs = smstream.FrameStream(eng, F, pipeline=2, result_sets=4)
            for _ in range(8):
                fs.run(d)
            fs.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                fs.run(d)
            fs.synchronize()
            ms = (time.perf_counter() - t) / 20 * 1e3
            nd = fs.results_host()["n_descriptors"]
            line.append("%s %.3f" % ("fork" if mode > 0 else "serial", ms))
            fs.close(); eng.close()
    print("dense fraction %.2f: %6d descriptors per frame (%.2e per input pixel): %s" % (p, nd // F, nd / F / (W * H), "  ".join(line)), flush=True)
    d.close()
