"""Throughput of every single-GPU BASELINE.json configuration (device-resident frames, packed results left in HBM),
one JSON line each.  bench.py's `value` is configs[2]; these are the DESIGN.md section 6 side numbers.
usage: python tools/bench_configs.py [--tile]   (--tile adds the 8192 x 8192 / 6 octaves case: ~40 GB of HBM)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
from tests.synth import blob_frame


def run(name, w, h, n_oct, frames, lockstep, reps, gray=False, kp_per_frame=32768, desc_per_frame=49152, depths=(2,)):
    dev = torch.device("cuda", 0)
    eng = sm.Engine(w, h, n_octaves=n_oct, max_batch=lockstep)
    fs = smstream.FrameStream(eng, frames, device=dev, kp_per_frame=kp_per_frame, desc_per_frame=desc_per_frame)
    base = [blob_frame(w, h, i, gray=gray) for i in range(min(frames, 8))]
    d = torch.from_numpy(np.stack([base[i % len(base)] for i in range(frames)])).to(dev)
    for _ in range(2):
        fs.run(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fs.run(d)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    r = fs.results_host()
    # the same with several steps in flight (FrameStream(pipeline=n): consecutive steps rotate over n contexts)
    flight = {}
    for depth in depths:
        fs2 = smstream.FrameStream(eng, frames, device=dev, kp_per_frame=kp_per_frame, desc_per_frame=desc_per_frame, pipeline=depth)
        for _ in range(2 * depth):
            fs2.run(d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(depth * reps):
            fs2.run(d)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / (depth * reps)
        r2 = fs2.results_host()
        assert (r2["n_keypoints"], r2["n_descriptors"]) == (r["n_keypoints"], r["n_descriptors"])
        flight["%d_in_flight" % depth] = {"ms_per_step": round(dt2 * 1e3, 3), "mpixels_per_s": round(frames * w * h / dt2 / 1e6, 1)}
        for e in fs2.engines[1:]:
            e.close()
        del fs2
    print(json.dumps({"config": name, "width": w, "height": h, "octaves": n_oct, "frames_per_step": frames, "lock_step": lockstep,
                      "ms_per_step": round(dt * 1e3, 3), "mpixels_per_s": round(frames * w * h / dt / 1e6, 1), **flight,
                      "keypoints": r["n_keypoints"], "descriptors": r["n_descriptors"]}), flush=True)
    eng.close()
    del fs, d
    torch.cuda.empty_cache()


if __name__ == "__main__":
    run("configs[0] shape on the GPU: 640x480 gray, 3 octaves, single frame", 640, 480, 3, 1, 1, 50, gray=True, depths=(2, 4))
    run("configs[1]: single 1920x1080 frame, 4 octaves", 1920, 1080, 4, 1, 1, 50, depths=(2, 4))
    run("configs[2]: 64 x 1920x1080, 4 octaves (bench.py value)", 1920, 1080, 4, 64, 64, 5)
    if "--tile" in sys.argv:
        run("configs[4]: single 8192x8192 tile, 6 octaves", 8192, 8192, 6, 1, 1, 3, kp_per_frame=1 << 20, desc_per_frame=3 << 19)
