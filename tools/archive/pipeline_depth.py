"""ms per 64 x 1920x1080 step against steps in flight (contexts alternating), device-resident, graph replay.
Run on the GPU box:  python tools/pipeline_depth.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if "--torch" in sys.argv:            # torch first: the process then runs on the HIP runtime torch bundles
    import torch
    torch.cuda.init()
import siftmetal_amd as sm
from siftmetal_amd import _capi, stream as smstream
from tests.synth import blob_frame
frames = np.stack([blob_frame(1920, 1080, i) for i in range(8)])
F = 64
eng = sm.Engine(1920, 1080, n_octaves=4, max_batch=F, graph_fork=int(os.environ.get("GRAPH_FORK", "0")))
if "--torch" in sys.argv:
    pass
d = smstream.DeviceFrames(np.concatenate([frames] * 8))
for pipe in [int(x) for x in os.environ.get("PIPES", "1,2,3,4").split(",")]:
    fs = smstream.FrameStream(eng, F, pipeline=pipe, result_sets=2 * pipe)
    for _ in range(4 * pipe):
        fs.run(d)
    fs.synchronize()
    res = []
    for rep in range(4):
        t = time.perf_counter()
        for _ in range(24):
            fs.run(d)
        fs.synchronize()
        res.append((time.perf_counter() - t) / 24 * 1e3)
    res.sort()
    print("pipeline %d: %.3f ms per step (median of 4 x 24)" % (pipe, res[1]), flush=True)
    fs.close()
