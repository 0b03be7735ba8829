"""Octave-0 blur launch times by layer (64 x 1080p) -- for SIFTMI_EXP_CHUNK_BIG sweeps with the experiment build.
usage: SIFTMI_LIB=tools/tmp_variants/libsiftmi_exp.so SIFTMI_EXP_CHUNK_BIG=<rows> python tools/chunk_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
import bench
F = 64
eng = sm.Engine(1920, 1080, n_octaves=4, max_batch=F)
fs = smstream.FrameStream(eng, F)
d = smstream.DeviceFrames(bench.make_frames(F, 8))
fs.run(d); fs.synchronize()
out = []
for rep in range(2):
    row = [round(eng.time_blur(0, l, 10) * 1e3, 1) for l in range(1, 6)]
    row1 = [round(eng.time_blur(1, l, 10) * 1e3, 1) for l in range(1, 6)]
    out.append((row, row1))
print("chunk big %s small %s: o0 layers 1-5 us per launch: %s  sum %.0f | o1: %s sum %.0f" % (os.environ.get("SIFTMI_EXP_CHUNK_BIG", "256"), os.environ.get("SIFTMI_EXP_CHUNK_SMALL", "160"), out[-1][0], sum(out[-1][0]), out[-1][1], sum(out[-1][1])), flush=True)
