"""Runs only the Gaussian-layer blur kernel (for rocprofv3 / PMC passes).
usage: python tools/prof_blur.py [octave] [layer] [iters] [batch] [W] [H]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import siftmetal_amd as sm
from tests.synth import blob_frame

o = int(sys.argv[1]) if len(sys.argv) > 1 else 0
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 5
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 8
W = int(sys.argv[5]) if len(sys.argv) > 5 else 1920
H = int(sys.argv[6]) if len(sys.argv) > 6 else 1080
eng = sm.Engine(W, H, n_octaves=4, max_batch=batch)
img = blob_frame(W, H, 0)
eng.detect_describe_batch(np.stack([img] * batch))
layers = range(1, 6) if layer == 0 else [layer]
for l in layers:
    ms = eng.time_blur(o, l, iters)
    gb = eng.blur_algorithmic_bytes(o) * batch / (ms * 1e-3) / 1e9
    print("octave %d layer %d taps %d batch %d: %.4f ms/launch  %.1f GB/s algorithmic" % (o, l, len(eng.weights(l)), batch, ms, gb))
