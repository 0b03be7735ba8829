"""Single 1920x1080 frame, one call at a time, for 1 ... 4 octaves: which chain of the forked launch graph bounds the call.
Run on the GPU box:  python tools/single_frame_octaves.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import siftmetal_amd as sm
from siftmetal_amd import _capi, stream as smstream
from tests.synth import blob_frame
L = _capi.load()
kw = {"blur_march_min_blocks": int(os.environ["SF_MARCH_MIN"])} if os.environ.get("SF_MARCH_MIN") else {}
for no in (1, 2, 3, 4):
    eng = sm.Engine(1920, 1080, n_octaves=no, max_batch=1, **kw)
    fs = smstream.FrameStream(eng, 1)
    d = smstream.DeviceFrames(blob_frame(1920, 1080, 0)[None])
    for _ in range(10):
        fs.run(d)
    L.siftmi_device_synchronize(0)
    res = []
    for rep in range(7):
        t = time.perf_counter()
        for _ in range(50):
            fs.run(d)
            fs.synchronize()
        res.append((time.perf_counter() - t) / 50 * 1e3)
    res.sort()
    r = fs.results_host()
    print("%d octaves: median %.3f ms (best %.3f), %d keypoints, %d descriptors" % (no, res[3], res[0], len(r["keypoints"]), len(r["descriptors"])), flush=True)
    fs.close(); eng.close()
