"""One 1080p frame per call (BASELINE configs[1]) repeated, for rocprofv3 --kernel-trace."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
from tests.synth import blob_frame

eng = sm.Engine(1920, 1080, n_octaves=4, max_batch=1)
fs = smstream.FrameStream(eng, 1)
d = smstream.DeviceFrames(blob_frame(1920, 1080, 0)[None])
for _ in range(12):
    fs.run(d)
    fs.synchronize()
print(fs.results_host()["n_descriptors"])
