"""(needs the SIFTMI_EXP_PHASE probe: `patch -p0 < tools/experiments/api_probes_r05.diff` before building the -DSIFTMI_EXPERIMENT variant)
Experiment (library built with -DSIFTMI_EXPERIMENT, SIFTMI_LIB pointing at it): the dense phase (seed, pyramid, extrema)
and the keypoint phase (refine ... pack) of a step as separate calls on separate streams, the keypoint stream restricted to
a subset of the CUs (hipExtStreamCreateWithCUMask), so that the VALU-bound keypoint kernels cannot take the whole chip from
the other context's HBM-bound dense kernels.  No torch.
usage: tools/build_variant.sh exp -DSIFTMI_EXPERIMENT; SIFTMI_LIB=$PWD/tools/tmp_variants/libsiftmi_exp.so python tools/cumask_probe.py [dense]
(round 5: re-run on DENSE frames, whose keypoint phase is half the step; complementary masks on both streams added)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SIFTMI_NO_GRAPH"] = "1"
import numpy as np

import siftmetal_amd as sm
from siftmetal_amd import _capi, stream as smstream
import bench

L = _capi.load()
hip = C.CDLL("libamdhip64.so.7")
dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
F, W, H = 64, 1920, 1080
frames = bench.make_dense_frames(F) if dense else bench.make_frames(F, 8)
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


def masked_stream(words):
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), arr)
    assert rc == 0, rc
    return s


class Ctx:
    def __init__(self):
        self.e = sm.Engine(W, H, n_octaves=4, nspo=3, max_batch=F)
        self.kp, self.ds = dev_alloc(KP * 44), dev_alloc(DS * 136)
        self.counts, self.totals = dev_alloc(2 * F * 4 * 4), dev_alloc(16)

    def call(self, stream, phase):
        os.environ["SIFTMI_EXP_PHASE"] = str(phase)
        self.e.detect_describe_batch_device(F, d.ptr, _capi.FMT_BGRA8, d.strides[1], d.strides[0], self.kp, KP, self.ds, DS, self.counts, self.totals, stream)

    def totals_host(self):
        t = np.zeros(4, np.int32)
        _capi.check(L.siftmi_memcpy(t.ctypes.data, self.totals, 16, 1))
        return t


def sync():
    _capi.check(L.siftmi_device_synchronize(0))


def timeit(step, n=16, warm=6):
    for _ in range(warm):
        step()
    sync()
    t = time.perf_counter()
    for _ in range(n):
        step()
    sync()
    return (time.perf_counter() - t) / n * 1e3


cs = [Ctx(), Ctx()]
sD, sB = plain_stream(), plain_stream()
print("one context, whole step per call (direct launches): %.3f ms" % timeit(lambda: cs[0].call(sD, 0)))
print("   dense phase alone: %.3f ms, keypoint phase alone: %.3f ms" % (timeit(lambda: cs[0].call(sD, 1)), timeit(lambda: cs[0].call(sD, 2))))
k = [0]


def alt():
    i = k[0] & 1
    k[0] += 1
    cs[i].call(sD if i == 0 else sB, 0)


print("two contexts, whole steps alternating on two streams: %.3f ms" % timeit(alt))
ref = [c.totals_host() for c in cs]
masks = {"all 256 CUs": [0xffffffff] * 8,
         "first 128 bits": [0xffffffff] * 4 + [0] * 4,
         "every 2nd bit (128)": [0x55555555] * 8, "3 of 4 bits (192)": [0x77777777] * 8, "every 4th bit (64)": [0x11111111] * 8,
         "5 of 8 bits (160)": [0x1f1f1f1f] * 8, "3 of 8 bits (96)": [0x07070707] * 8}
for name, words in masks.items():
    sK = masked_stream(words)
    print("   keypoint phase alone on [%s]: %.3f ms" % (name, timeit(lambda: cs[0].call(sK, 2), n=8, warm=3)))

    def phased():
        i = k[0] & 1
        k[0] += 1
        cs[i].call(sD, 1)            # dense phases of both contexts back to back on one stream
        cs[i].call(sK, 2)            # keypoint phases on the masked one; the library orders a context's calls among themselves
    print("two contexts, dense phases on one stream / keypoint phases on a stream with [%s]: %.3f ms" % (name, timeit(phased)))
    sync()
    assert all((c.totals_host() == r).all() for c, r in zip(cs, ref))
    # the dense phases confined to the COMPLEMENT of the keypoint stream's CUs (no sharing at all)
    comp = [(~wd) & 0xffffffff for wd in words]
    if any(comp):
        sDm = masked_stream(comp)

        def phased2():
            i = k[0] & 1
            k[0] += 1
            cs[i].call(sDm, 1)
            cs[i].call(sK, 2)
        print("two contexts, dense phases on the complement / keypoint phases on [%s]: %.3f ms" % (name, timeit(phased2)), flush=True)
        sync()
        assert all((c.totals_host() == r).all() for c, r in zip(cs, ref))
        hip.hipStreamDestroy(sDm)
    hip.hipStreamDestroy(sK)
