"""One 8192 x 8192 / 6-octave tile per call (BASELINE configs[4]) repeated, for rocprofv3 --kernel-trace."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
from tests.synth import blob_frame

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
eng = sm.Engine(W, W, n_octaves=6, max_batch=1)
fs = smstream.FrameStream(eng, 1, device=dev, kp_per_frame=1 << 20, desc_per_frame=3 << 19)
d = torch.from_numpy(blob_frame(W, W, 0)[None]).to(dev)
for _ in range(4):
    fs.run(d)
torch.cuda.synchronize()
print(fs.results_host()["n_descriptors"])
