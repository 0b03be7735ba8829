"""Does running two half-batches on two contexts / streams overlap the VALU-bound keypoint stages of one with the HBM-bound
dense stages of the other?  usage: python tools/overlap_experiment.py [dense]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as ge
ge.build()
import siftmetal_amd as sm
from siftmetal_amd import stream as smstream
import bench

dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
F, W, H = 64, 1920, 1080
dev = torch.device("cuda", 0)
frames = bench.make_dense_frames(F) if dense else bench.make_frames(F, 8)
d = torch.from_numpy(frames).to(dev)

def timeit(step, n=8, warm=4):
    for _ in range(warm): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

e = sm.Engine(W, H, n_octaves=4, nspo=3, max_batch=64)
r = smstream.FrameStream(e, F, device=dev)
print("one context, lock-step 64: %.3f ms" % timeit(lambda: r.run(d)))
del r; e.close()
for parts in (2, 4):
    n = F // parts
    es = [sm.Engine(W, H, n_octaves=4, nspo=3, max_batch=n) for _ in range(parts)]
    rs = [smstream.FrameStream(x, n, device=dev) for x in es]
    ds = [d[i * n:(i + 1) * n] for i in range(parts)]
    ss = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    def step():                      # each context under its own current stream: FrameStream orders its launch stream with that one only
        for q, x, st in zip(rs, ds, ss):
            with torch.cuda.stream(st): q.run(x)
    print("%d contexts x %d frames on %d streams: %.3f ms" % (parts, n, parts, timeit(step)))
    def step_seq():
        for q, x in zip(rs, ds):
            q.run(x); torch.cuda.current_stream().synchronize()
    print("   the same, one after the other: %.3f ms" % timeit(step_seq))
    del rs
    for x in es: x.close()

# consecutive 64-frame steps alternating between two contexts (two pyramid sets) on two streams: step k+1's dense stages can
# run under step k's keypoint stages
es = [sm.Engine(W, H, n_octaves=4, nspo=3, max_batch=64) for _ in range(2)]
rs = [smstream.FrameStream(x, F, device=dev) for x in es]
ss = [torch.cuda.Stream(device=dev) for _ in range(2)]
k = [0]
def step_alt():
    i = k[0] & 1; k[0] += 1
    with torch.cuda.stream(ss[i]): rs[i].run(d)
print("alternating 64-frame steps over 2 contexts / 2 streams: %.3f ms per step" % timeit(step_alt, n=16, warm=6))
for x in es: x.close()
for depth in (3, 4):
    es = [sm.Engine(W, H, n_octaves=4, nspo=3, max_batch=64) for _ in range(depth)]
    rs = [smstream.FrameStream(x, F, device=dev) for x in es]
    ss = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    k = [0]
    def step_alt():
        i = k[0] % depth; k[0] += 1
        with torch.cuda.stream(ss[i]): rs[i].run(d)
    print("alternating 64-frame steps over %d contexts / streams: %.3f ms per step" % (depth, timeit(step_alt, n=24, warm=12)))
    del rs
    for x in es: x.close()
