"""Where the host-fed stream's time goes: upload alone, host-fed steps without reading results, with results (copy / views).
    python tools/host_io_probe.py [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--torch" in sys.argv:            # as bench.py: torch (and its bundled HIP runtime) loaded and initialised first
    sys.argv.remove("--torch")
    import torch
    torch.cuda.init()
    torch.cuda.synchronize()
    print("torch loaded first:", torch.__version__)
import __graft_entry__ as ge  # noqa: E402

ge.build()
import siftmetal_amd as sm  # noqa: E402
from siftmetal_amd import _capi, stream as smstream  # noqa: E402
from tests.synth import blob_frame  # noqa: E402

W, H, F = 1920, 1080, 64
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = _capi.load()
frames = np.stack([blob_frame(W, H, i % 8) for i in range(F)])
pin = sm.pinned_empty(frames.shape, np.uint8)
pin[...] = frames


def sync():
    _capi.check(L.siftmi_device_synchronize(0))


def timed(fn, n=steps, warm=10):
    for _ in range(warm):
        fn()
    sync()
    t = time.perf_counter()
    for i in range(n):
        fn()
    sync()
    return (time.perf_counter() - t) / n * 1e3


d = smstream.DeviceFrames(frames)
t = time.perf_counter()
for _ in range(5):
    _capi.check(L.siftmi_memcpy(d.ptr, pin.ctypes.data, pin.nbytes, 0))
print("synchronous H2D of one step: %.2f ms (%.1f GB/s)" % ((time.perf_counter() - t) / 5 * 1e3, pin.nbytes * 5 / (time.perf_counter() - t) / 1e9))

for pipeline, sets in ((2, 4), (2, 6), (1, 2)):
    eng = sm.Engine(W, H, n_octaves=4, max_batch=F)
    fs = smstream.FrameStream(eng, F, pipeline=pipeline, result_sets=sets)
    print("pipeline %d, %d result sets" % (pipeline, sets))
    print("  resident steps:                   %.3f ms" % timed(lambda: fs.run(d)))
    print("  host-fed, results never read:     %.3f ms" % timed(lambda: fs.run_host(pin)))
    back = pipeline

    def step_read(copy):
        fs.run_host(pin)
        if fs.step_no >= back:
            fs.results_host(back=back, copy=copy)
    print("  host-fed, results read (views):   %.3f ms" % timed(lambda: step_read(False)))
    print("  host-fed, results read (copies):  %.3f ms" % timed(lambda: step_read(True)))

    def step_dev():
        fs.run(d)
        if fs.step_no >= back:
            fs.results_host(back=back, copy=False)
    print("  resident, results read (views):   %.3f ms" % timed(step_dev))
    fs.close()
    eng.close()
