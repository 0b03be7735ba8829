import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if not os.environ.get("SIFTMI_LIB"):
    import __graft_entry__ as ge
    ge.build()
import siftmetal_amd as sm
from siftmetal_amd import _capi, stream as smstream
from tests.synth import blob_frame
L = _capi.load()
for (w, h, no) in ((1920, 1080, 4), (640, 480, 3)):
    eng = sm.Engine(w, h, n_octaves=no, max_batch=1, **({"blur_march_min_blocks": int(os.environ["SF_MARCH_MIN"])} if os.environ.get("SF_MARCH_MIN") else {}),
                    **({"blur_chain_max_tiles": int(os.environ["SF_CHAIN_MAX"])} if os.environ.get("SF_CHAIN_MAX") else {}))
    fs = smstream.FrameStream(eng, 1)
    d = smstream.DeviceFrames(blob_frame(w, h, 0)[None])
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
    # back-to-back (no host sync between calls: the next call's launch overlaps the tail of this one)
    t = time.perf_counter()
    for _ in range(200):
        fs.run(d)
    L.siftmi_device_synchronize(0)
    b2b = (time.perf_counter() - t) / 200 * 1e3
    print("%dx%d: one call at a time median %.3f ms (best %.3f); back to back %.3f ms" % (w, h, res[3], res[0], b2b))
    fs.close(); eng.close()
