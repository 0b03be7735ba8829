"""One 1080p frame per call (BASELINE configs[1]), frames resident: ms per call from an idle queue (synchronise after every call) and with
four calls queued.  usage: [SIFTMI_LIB=<build>] python tools/single_frame_time.py [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
from tests.synth import blob_frame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
eng = sm.Engine(1920, 1080, n_octaves=4, max_batch=1)
fs = smstream.FrameStream(eng, 1)
d = smstream.DeviceFrames(blob_frame(1920, 1080, 0)[None])
for _ in range(20):
    fs.run(d)
    fs.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(n):
        fs.run(d)
        fs.synchronize()
    best = min(best, (time.perf_counter() - t0) / n)
t0 = time.perf_counter()
for _ in range(n):
    fs.run(d)
fs.synchronize()
q = (time.perf_counter() - t0) / n
print("%s coop_wg=%s: %.1f us per frame (synchronised calls, best of 5 x %d), %.1f us queued; %d descriptors" %
      (os.environ.get("SIFTMI_LIB", "libsiftmi.so"), os.environ.get("SIFTMI_EXP_COOP_WG", "1024"), best * 1e6, n, q * 1e6, fs.results_host()["n_descriptors"]), flush=True)
