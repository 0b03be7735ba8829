"""Soak: a few thousand pipelined steps (resident and host-fed, alternating inputs), results checked against the first pass,
device memory watched for growth.  usage: python tools/soak.py [steps]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
from tests.synth import blob_frame
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
F, W, H = 8, 1280, 720
sets = [np.stack([blob_frame(W, H, 10 * j + i, n_blobs=200 + 150 * j) for i in range(F)]) for j in range(3)]
eng = sm.Engine(W, H, n_octaves=4, max_batch=F)
want = [eng.detect_describe_batch(f) for f in sets]
fs = smstream.FrameStream(eng, F, device=dev, pipeline=2, result_sets=4)
dsets = [torch.from_numpy(f).to(dev) for f in sets]
pins = [torch.from_numpy(f).pin_memory() for f in sets]
def check(r, j, step):
    k, kc, d, dc = want[j]
    assert r["keypoints"].tobytes() == k.tobytes() and r["descriptors"].tobytes() == d.tobytes(), step
free0 = None
t0 = time.time()
for step in range(steps):
    j = step % 3
    if (step // 50) % 2 == 0: fs.run(dsets[j])
    else: fs.run_host(pins[j])
    if step >= 2: check(fs.results_host(back=2), (step - 2) % 3, step - 2)
    if step == 200: free0 = torch.cuda.mem_get_info()[0]
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("%d steps in %.1f s, all results byte-identical; free device memory after step 200: %.1f MB, at the end: %.1f MB" % (steps, time.time() - t0, free0 / 1e6, free1 / 1e6))
