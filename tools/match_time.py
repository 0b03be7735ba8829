import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
eng = sm.Engine(64, 64, n_octaves=1)
rng = np.random.default_rng(0)
for (ns, nt) in [(2500, 2300), (8000, 8000), (20000, 20000), (40000, 40000)]:
    tgt = np.zeros(nt, sm.descriptor_dtype); tgt["features"] = np.clip(np.abs(rng.normal(0, 40, (nt, 128))), 0, 255)
    src = np.zeros(ns, sm.descriptor_dtype); pick = rng.integers(0, nt, ns)
    src["features"] = np.clip(tgt["features"][pick].astype(np.int32) + rng.integers(-10, 11, (ns, 128)), 0, 255)
    d_src, d_tgt = smstream.DeviceFrames(src.view(np.uint8)), smstream.DeviceFrames(tgt.view(np.uint8))
    out = smstream.DeviceFrames(np.zeros(ns * 12 + 4, np.uint8))
    for _ in range(5): eng.match_device(d_src.ptr, ns, d_tgt.ptr, nt, out.ptr + 4, out.ptr)
    eng.synchronize()
    for reps in (1, 50):
        t0 = time.perf_counter()
        for _ in range(reps): eng.match_device(d_src.ptr, ns, d_tgt.ptr, nt, out.ptr + 4, out.ptr)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%s %d x %d: %.1f us per call (%d queued)  = %.3f of 5 POP/s" % (os.environ.get("SIFTMI_MATCH_NO_FUSE", "fused"), ns, nt, dt * 1e6, reps, ns * nt * 256 / dt / 5e15), flush=True)
