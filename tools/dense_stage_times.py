"""Stage times of one 64 x 1080p step on dense natural texture (bench.py's config.dense) or on the benchmark frames, per-launch
hipEvents (serialised launches: no overlap between octave chains).
The line ends with a digest of the packed keypoint + descriptor records: equal digests from two builds = byte-identical results.
usage: [SIFTMI_LIB=<experiment build>] python tools/dense_stage_times.py [steps] [dense|bench] [patch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
F = 64
patch = 1 if (len(sys.argv) > 3 and sys.argv[3] == "patch") else 0      # siftmi_config.descriptor_patch_lds
import json
extra = json.loads(os.environ.get("SIFTMI_ENGINE_KW", "{}"))                # e.g. SIFTMI_ENGINE_KW='{"blur_chain_max_tiles": 4096}'
eng = sm.Engine(1920, 1080, n_octaves=4, max_batch=F, descriptor_patch_lds=patch, **extra)
fs = smstream.FrameStream(eng, F)
kind = sys.argv[2] if len(sys.argv) > 2 else "dense"
d = smstream.DeviceFrames(bench.make_dense_frames(F) if kind == "dense" else bench.make_frames(F, 16))
for _ in range(2):
    fs.run(d)
fs.synchronize()
eng.enable_timings(True)
eng.reset_timings()
for _ in range(steps):
    fs.run(d)
fs.synchronize()
tm = eng.timings()
r = fs.results_host()
import hashlib
import numpy as np
digest = hashlib.sha256(np.ascontiguousarray(r["keypoints"]).tobytes() + np.ascontiguousarray(r["descriptors"]).tobytes()).hexdigest()[:16]
print(os.environ.get("SIFTMI_LIB", "libsiftmi.so") + (" descriptor_patch_lds=1" if patch else "") + (" " + json.dumps(extra) if extra else ""), {k: round(v[0] / steps, 3) for k, v in tm.items() if v[0] > 0}, r["n_keypoints"], r["n_descriptors"],
      "sha256 of the packed records", digest, flush=True)
