"""Runs the whole detect+describe path on a few synthetic 1080p frames (for rocprofv3).
usage: python tools/prof_pipeline.py [frames] [batch] [reps] [dense]   (dense: mirror-tiled butterfly frames instead of blob fields)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import siftmetal_amd as sm
from tests.synth import blob_frame

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if len(sys.argv) > 4 and sys.argv[4] == "dense":
    sys.argv = sys.argv[:1]
    import bench
    frames = bench.make_dense_frames(F)
else:
    frames = np.stack([blob_frame(1920, 1080, i % 8) for i in range(F)])
eng = sm.Engine(1920, 1080, n_octaves=4, max_batch=B)
for _ in range(reps):
    k, kc, d, dc = eng.detect_describe_batch(frames)
print("frames", F, "keypoints", len(k), "descriptors", len(d))
