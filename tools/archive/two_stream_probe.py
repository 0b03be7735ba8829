"""Upper-bound probe: two independent engines on two HIP streams, each taking half of the 64 frames,
vs one engine taking all 64 (same lock-step batch)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
from tests.synth import blob_frame

dev = torch.device("cuda", 0)
base = [blob_frame(1920, 1080, i) for i in range(8)]
frames = torch.from_numpy(np.stack([base[i % 8] for i in range(64)])).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def bench(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


e0 = sm.Engine(1920, 1080, n_octaves=4, max_batch=B)
f0 = smstream.FrameStream(e0, 64, device=dev)
t_single = bench(lambda: f0.run(frames))
del f0, e0
e1 = sm.Engine(1920, 1080, n_octaves=4, max_batch=B)
e2 = sm.Engine(1920, 1080, n_octaves=4, max_batch=B)
f1 = smstream.FrameStream(e1, 32, device=dev)
f2 = smstream.FrameStream(e2, 32, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h1, h2 = frames[:32].contiguous(), frames[32:].contiguous()


def both():
    with torch.cuda.stream(s1):
        f1.run(h1)
    with torch.cuda.stream(s2):
        f2.run(h2)


t_two = bench(both)
print("lock-step batch %d: one engine/stream %.2f ms per 64 frames; two engines on two streams %.2f ms (%.1f %% faster)" %
      (B, t_single, t_two, 100 * (t_single / t_two - 1)))
