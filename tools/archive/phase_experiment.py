"""Experiment (needs a library built with -DSIFTMI_EXPERIMENT): the dense phase (seed, pyramid, extrema) and the keypoint
phase (refine ... pack) of a step as separate calls, dense phases of alternating contexts back to back on one stream and
keypoint phases on another, so that a keypoint phase always runs under the OTHER context's dense phase.
usage: SIFTMI_NO_GRAPH=1 python tools/phase_experiment.py [dense]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
os.environ["SIFTMI_NO_GRAPH"] = "1"
import siftmetal_amd as sm
from siftmetal_amd import _capi, dist as smdist
import bench

dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
F, W, H = 64, 1920, 1080
dev = torch.device("cuda", 0)
frames = bench.make_dense_frames(F) if dense else bench.make_frames(F, 8)
d = torch.from_numpy(frames).to(dev)
KP, DS = 32768 * F, 49152 * F

class Ctx:
    def __init__(self):
        self.e = sm.Engine(W, H, n_octaves=4, nspo=3, max_batch=F)
        self.kp = torch.empty(KP * 44, dtype=torch.uint8, device=dev)
        self.ds = torch.empty(DS * 136, dtype=torch.uint8, device=dev)
        self.counts = torch.zeros((2, F, 4), dtype=torch.int32, device=dev)
        self.totals = torch.zeros(4, dtype=torch.int32, device=dev)
    def call(self, stream, phase):
        os.environ["SIFTMI_EXP_PHASE"] = str(phase)
        self.e.detect_describe_batch_device(F, d.data_ptr(), _capi.FMT_BGRA8, d.stride(1), d.stride(0), self.kp.data_ptr(), KP,
                                            self.ds.data_ptr(), DS, self.counts.data_ptr(), self.totals.data_ptr(), stream.cuda_stream)

def timeit(step, n=16, warm=6):
    for _ in range(warm): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

cs = [Ctx(), Ctx()]
sD, sK, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
print("one context, whole step per call (direct launches): %.3f ms" % timeit(lambda: cs[0].call(sD, 0)))
print("   dense phase alone: %.3f ms, keypoint phase alone: %.3f ms" % (timeit(lambda: cs[0].call(sD, 1)), timeit(lambda: cs[0].call(sD, 2))))
k = [0]
def alt():
    i = k[0] & 1; k[0] += 1
    cs[i].call(sD if i == 0 else sB, 0)
print("two contexts, whole steps alternating on two streams: %.3f ms" % timeit(alt))
ref = [c.totals.cpu().numpy().copy() for c in cs]
def phased():
    i = k[0] & 1; k[0] += 1
    cs[i].call(sD, 1)            # dense phases of both contexts back to back on one stream
    cs[i].call(sK, 2)            # keypoint phases on the other; the library orders a context's calls among themselves
print("two contexts, dense phases on one stream / keypoint phases on another: %.3f ms" % timeit(phased))
torch.cuda.synchronize()
print("totals unchanged:", [bool((c.totals.cpu().numpy() == r).all()) for c, r in zip(cs, ref)], ref[0][:2])
