"""Per-layer blur launch times of the pipeline (with its DEC / ACT outputs) for every octave: python tools/blur_layers.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import siftmetal_amd as sm
from tests.synth import blob_frame

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
noact = len(sys.argv) > 2 and sys.argv[2] == "noact"          # count_raw_extrema = 1: no activity flags -> the plain kernels
eng = sm.Engine(1920, 1080, n_octaves=4, max_batch=B, count_raw_extrema=1 if noact else 0)
frames = np.stack([blob_frame(1920, 1080, i % 8) for i in range(B)])
eng.detect_describe_batch(frames)
for o in range(4):
    row = []
    for layer in range(1, 6):
        ms = eng.time_blur(o, layer, 10)
        row.append("L%d R=%2d %7.1f us %6.0f GB/s" % (layer, len(eng.weights(layer)) // 2, ms * 1e3, eng.blur_algorithmic_bytes(o) * B / (ms * 1e-3) / 1e9))
    print("octave %d (batch %d%s): " % (o, B, ", no activity flags" if noact else "") + " | ".join(row), flush=True)
