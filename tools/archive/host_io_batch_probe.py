"""config.host_io of bench.py alone: 64 x 1080p BGRA8 frames in pinned host memory through the synchronous siftmi_detect_describe_batch
(sub-batches of max_batch frames), packed results copied back.  usage: python tools/host_io_batch_probe.py [max_batch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import siftmetal_amd as sm
import bench
F, W, H = 64, 1920, 1080
frames = bench.make_frames(F, 64)
pin = sm.pinned_empty(frames.shape, np.uint8)
pin[...] = frames
for mb in [int(a) for a in sys.argv[1:]] or [16]:
    eng = sm.Engine(W, H, n_octaves=4, max_batch=mb, graph_fork=int(os.environ.get('FORK', '0')))
    eng.detect_describe_batch(pin, copy=False)
    eng.detect_describe_batch(pin, copy=False)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        k, kc, d, dc = eng.detect_describe_batch(pin, copy=False)
        ts.append((time.perf_counter() - t) * 1e3)
    print("max_batch %2d: %.3f ms per 64-frame call (min %.3f)  %d keypoints %d descriptors" % (mb, sorted(ts)[2], min(ts), len(k), len(d)), flush=True)
    # the same call on frames already resident in HBM (on_device = 1): the sub-batched launch sequence and the copy-back alone
    import ctypes as C
    from siftmetal_amd import _capi, stream as smstream
    dfr = smstream.DeviceFrames(frames)
    outs = [C.c_void_p() for _ in range(4)]
    def dev_call():
        _capi.check(eng.L.siftmi_detect_describe_batch(eng.h, F, dfr.ptr, _capi.FMT_BGRA8, W * 4, W * H * 4, 1, *[C.byref(o) for o in outs]))
    dev_call(); dev_call()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); dev_call(); ts.append((time.perf_counter() - t) * 1e3)
    print("             frames resident in HBM: %.3f ms (min %.3f)" % (sorted(ts)[2], min(ts)), flush=True)
    dfr.close()
    eng.close()
