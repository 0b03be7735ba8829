"""Host-fed frame stream (bench.py's headline): how far behind the submits the results are read, and how many result sets / staging
buffers rotate.  The upload of a step is 9.2 ms of PCIe, a step's kernels 9.4 ms of GPU: the two chains only overlap fully while the
copy stream always has the next upload queued, i.e. while the host submits step k+1 before upload k has finished -- and the host
blocks in siftmi_stream_result_host(back) until step k - back is done.
    python tools/host_fed_depth_probe.py [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import siftmetal_amd as sm  # noqa: E402
from siftmetal_amd import _capi, stream as smstream  # noqa: E402
from tests.synth import blob_frame  # noqa: E402
import bench  # noqa: E402

W, H, F = 1920, 1080, 64
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kind = sys.argv[2] if len(sys.argv) > 2 else "bench"
L = _capi.load()
frames = bench.make_dense_frames(F) if kind == "dense" else bench.make_frames(F, 64)
pin = sm.pinned_empty(frames.shape, np.uint8)
pin[...] = frames


def sync():
    _capi.check(L.siftmi_device_synchronize(0))


CASES = ((2, 4, 2), (2, 4, 3), (2, 6, 3), (1, 4, 2)) if os.environ.get('PROBE_SHORT') else ((2, 4, 2), (2, 4, 3), (2, 6, 3), (2, 6, 4), (2, 8, 5), (1, 4, 2), (1, 4, 3), (3, 6, 4))
for pipeline, sets, back in CASES:
    eng = sm.Engine(W, H, n_octaves=4, max_batch=F)
    fs = smstream.FrameStream(eng, F, pipeline=pipeline, result_sets=sets)

    def step():
        fs.run_host(pin)
        if fs.step_no >= back:
            fs.results_host(back=back, copy=False)

    for _ in range(2 * fs.n_sets + 2):
        step()
    sync()
    t = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    ms = (time.perf_counter() - t) / steps * 1e3
    print("upload streams %s; %s frames: %d contexts, %d result sets, results read %d steps late: %.3f ms/step" % (os.environ.get("SIFTMI_UPLOAD_STREAMS", "default"), kind, pipeline, fs.n_sets, back, ms), flush=True)
    fs.close(); eng.close()
