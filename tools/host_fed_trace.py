"""A short host-fed stream run for a timeline (rocprofv3 --kernel-trace --memory-copy-trace): 16 steps, results read 2 steps late.
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/host_fed_trace.py [resident]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import siftmetal_amd as sm  # noqa: E402
from siftmetal_amd import _capi, stream as smstream  # noqa: E402
import bench  # noqa: E402

W, H, F = 1920, 1080, 64
resident = len(sys.argv) > 1 and sys.argv[1] == "resident"
L = _capi.load()
frames = bench.make_frames(F, 16)
pin = sm.pinned_empty(frames.shape, np.uint8)
pin[...] = frames
d = smstream.DeviceFrames(frames)
eng = sm.Engine(W, H, n_octaves=4, max_batch=F)
fs = smstream.FrameStream(eng, F, pipeline=2, result_sets=4)
back = 2


def step():
    if resident:
        fs.run(d)
    else:
        fs.run_host(pin)
    if fs.step_no >= back:
        fs.results_host(back=back, copy=False)


for _ in range(10):
    step()
_capi.check(L.siftmi_device_synchronize(0))
t = time.perf_counter()
for _ in range(16):
    step()
_capi.check(L.siftmi_device_synchronize(0))
print("%s: %.3f ms/step" % ("resident" if resident else "host-fed", (time.perf_counter() - t) / 16 * 1e3), flush=True)
