"""Two steps in flight with the two launch streams at different priorities.  usage: python tools/priority_experiment.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
import bench
F, W, H = 64, 1920, 1080
dev = torch.device("cuda", 0)
d = torch.from_numpy(bench.make_frames(F, 8)).to(dev)
def timeit(step, n=20, warm=10):
    for _ in range(warm): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
eng = sm.Engine(W, H, n_octaves=4, nspo=3, max_batch=F)
for pr in [(0, 0), (-1, 0), (-1, -1)]:
    fs = smstream.FrameStream(eng, F, device=dev, pipeline=2)
    fs.launch_streams = [torch.cuda.Stream(device=dev, priority=p) for p in pr]
    print("launch stream priorities", pr, ": %.3f ms per step" % timeit(lambda: fs.run(d)))
    fs.close()
