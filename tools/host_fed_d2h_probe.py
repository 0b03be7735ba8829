"""Host-fed frame stream: does the copy of every step's results to host memory (27 MB on the benchmark frames, on the copy engines) stand
in the way of the uploads (531 MB per step, the chain that bounds the host-fed step)?  Three loops on the same stream shape as bench.py:
results read on the host / results only waited for on the device / nothing read.
usage: python tools/host_fed_d2h_probe.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import siftmetal_amd as sm
from siftmetal_amd import _capi, stream as smstream
import bench

W, H, F = 1920, 1080, 64
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = _capi.load()
frames = bench.make_frames(F, 64)
pin = sm.pinned_empty(frames.shape, np.uint8)
pin[...] = frames


def sync():
    _capi.check(L.siftmi_device_synchronize(0))


for mode in ("host results (bench.py)", "device results only", "no result read", "host results (bench.py)"):
    eng = sm.Engine(W, H, n_octaves=4, max_batch=F)
    fs = smstream.FrameStream(eng, F, pipeline=2, result_sets=4)

    def step():
        fs.run_host(pin)
        if fs.step_no >= 2:
            if mode.startswith("host"):
                fs.results_host(back=2, copy=False)
            elif mode.startswith("device"):
                fs.result_device(2)

    for _ in range(10):
        step()
    sync()
    t = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    print("%-26s %.3f ms/step" % (mode, (time.perf_counter() - t) / steps * 1e3), flush=True)
    fs.close(); eng.close()
