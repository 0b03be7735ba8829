"""(needs the SIFTMI_EXP_PHASE probe: `patch -p0 < tools/experiments/api_probes_r05.diff` before building the -DSIFTMI_EXPERIMENT variant)
Experiment (library built with -DSIFTMI_EXPERIMENT, SIFTMI_LIB pointing at it): can the vector-bound keypoint phase of step k
(refine ... pack) run INSIDE the load/store-bound dense phase (seed, pyramid, scan) of step k + 1?  Two large kernels on two hardware
queues take turns (profiles/cumask_dense_r05.log: the strict two-stream schedule costs the SUM of the phases), so here the orientation
and descriptor kernels are launched as a FEW wavefronts per CU over all (frame, octave) groups (SIFTMI_EXP_KP_FLAT = wavefronts per CU;
keypoint_kernels.hip.h, FLAT) -- a grid that is resident at once and leaves the rest of every CU to the other stream's blur workgroups --
and the blur launches can be held to fewer workgroups per CU by unused LDS (SIFTMI_EXP_RING_PAD_LDS).
usage (the FLAT launch form is not in the tree: git apply tools/experiments/phase_mix_r05.diff first): tools/build_variant.sh exp -DSIFTMI_EXPERIMENT;
SIFTMI_LIB=$PWD/tools/tmp_variants/libsiftmi_exp.so python tools/experiments/phase_mix_probe.py [dense|bench]"""
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SIFTMI_NO_GRAPH"] = "1"
import numpy as np

import siftmetal_amd as sm
from siftmetal_amd import _capi, stream as smstream
import bench

L = _capi.load()
hip = C.CDLL("libamdhip64.so.7")
kind = sys.argv[1] if len(sys.argv) > 1 else "dense"
F, W, H = 64, 1920, 1080
frames = bench.make_dense_frames(F) if kind == "dense" else bench.make_frames(F, 16)
d = smstream.DeviceFrames(frames)
KP, DS = 32768 * F, 49152 * F


def dev_alloc(n):
    p = C.c_void_p()
    _capi.check(L.siftmi_device_alloc(0, n, C.byref(p)))
    return p.value


def plain_stream():
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    return s


class Ctx:
    def __init__(self):
        self.e = sm.Engine(W, H, n_octaves=4, nspo=3, max_batch=F)
        self.kp, self.ds = dev_alloc(KP * 44), dev_alloc(DS * 136)
        self.counts, self.totals = dev_alloc(2 * F * 4 * 4), dev_alloc(16)

    def call(self, stream, phase, flat=0, pad=0):
        os.environ["SIFTMI_EXP_PHASE"] = str(phase)
        os.environ["SIFTMI_EXP_KP_FLAT"] = str(flat)
        os.environ["SIFTMI_EXP_RING_PAD_LDS"] = str(pad)
        self.e.detect_describe_batch_device(F, d.ptr, _capi.FMT_BGRA8, d.strides[1], d.strides[0], self.kp, KP, self.ds, DS, self.counts, self.totals, stream)

    def digest(self):
        t = np.zeros(4, np.int32)
        _capi.check(L.siftmi_memcpy(t.ctypes.data, self.totals, 16, 1))
        k = np.zeros(int(t[0]) * 44, np.uint8)
        ds = np.zeros(int(t[1]) * 136, np.uint8)
        _capi.check(L.siftmi_memcpy(k.ctypes.data, self.kp, k.nbytes, 1))
        _capi.check(L.siftmi_memcpy(ds.ctypes.data, self.ds, ds.nbytes, 1))
        return (int(t[0]), int(t[1]), hashlib.sha256(k.tobytes() + ds.tobytes()).hexdigest()[:16])


def sync():
    _capi.check(L.siftmi_device_synchronize(0))


def timeit(step, n=12, warm=4):
    for _ in range(warm):
        step()
    sync()
    t = time.perf_counter()
    for _ in range(n):
        step()
    sync()
    return (time.perf_counter() - t) / n * 1e3


cs = [Ctx(), Ctx()]
sD, sK, sB = plain_stream(), plain_stream(), plain_stream()
print("%s frames; one context, whole step per call (direct launches): %.3f ms" % (kind, timeit(lambda: cs[0].call(sD, 0))))
sync()
ref = cs[0].digest()
print("   dense phase alone: %.3f ms, keypoint phase alone: %.3f ms; records %s" % (timeit(lambda: cs[0].call(sD, 1)), timeit(lambda: cs[0].call(sD, 2)), ref))
k = [0]


def alt():
    i = k[0] & 1
    k[0] += 1
    cs[i].call(sD if i == 0 else sB, 0)


print("two contexts, whole steps alternating on two streams (what ships): %.3f ms" % timeit(alt), flush=True)
for flat in (0, 4, 8, 12, 16):
    if flat:
        print("   keypoint phase alone, %2d wavefronts per CU: %.3f ms" % (flat, timeit(lambda: cs[0].call(sK, 2, flat), n=6, warm=2)))
    for pad in (0, 14336, 43008):          # 4 / 3 / 2 ring workgroups per CU

        def phased():
            i = k[0] & 1
            k[0] += 1
            cs[i].call(sD, 1, flat, pad)   # dense phases of both contexts back to back on one stream
            cs[i].call(sK, 2, flat, pad)   # keypoint phases on the other; the library orders a context's calls among themselves
        ms = timeit(phased)
        sync()
        ok = cs[0].digest() == ref and cs[1].digest() == ref
        print("two contexts, dense phases on one stream / keypoint phases on another; keypoint kernels %s, ring LDS pad %5d B: %.3f ms  records %s" %
              ("%2d wavefronts per CU" % flat if flat else "as shipped        ", pad, ms, "identical" if ok else "DIFFER"), flush=True)
