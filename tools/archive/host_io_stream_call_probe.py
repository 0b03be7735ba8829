"""What a SYNCHRONOUS 64 x 1080p call would cost if it drove the frame stream (two contexts, rotating staging, copy-back per step) from
a standing start: submit the frames as 64 / F steps of F frames, read every step's results on the host, synchronise; repeated from idle.
Compare with config.host_io (siftmi_detect_describe_batch, one context).  usage: python tools/host_io_stream_call_probe.py [F ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
import bench
N, W, H = 64, 1920, 1080
frames = bench.make_frames(N, 64)
pin = sm.pinned_empty(frames.shape, np.uint8)
pin[...] = frames
for F in [int(a) for a in sys.argv[1:]] or [16]:
    for pipe in (1, 2):
        eng = sm.Engine(W, H, n_octaves=4, max_batch=F)
        n_steps = N // F
        fs = smstream.FrameStream(eng, F, pipeline=pipe, result_sets=max(n_steps, pipe))
        def call():
            for i in range(n_steps):
                fs.run_host(pin[i * F:(i + 1) * F])
            tot = [0, 0]
            for b in range(n_steps - 1, -1, -1):
                r = fs.results_host(back=b, copy=False)
                tot[0] += r["n_keypoints"]; tot[1] += r["n_descriptors"]
            fs.synchronize()
            return tot
        call(); call()
        ts = []
        for _ in range(5):
            t = time.perf_counter(); tot = call(); ts.append((time.perf_counter() - t) * 1e3)
        print("F %2d, %d context(s): %.3f ms per 64-frame call from idle (min %.3f)  %d keypoints %d descriptors" % (F, pipe, sorted(ts)[2], min(ts), tot[0], tot[1]), flush=True)
        fs.close(); eng.close()
