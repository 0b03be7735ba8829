"""siftmetal_amd -- MI355X-native SIFT detect+describe behind SIFTMetal's API shape.

Host-side mirror of the reference's public interface (names, argument meaning, result types):
    SIFT(device:configuration:)            Sources/SIFTMetal/SIFT/SIFT.swift:112-143
    SIFT.getKeypoints(_:)                  SIFT.swift:147-152   -> [[SIFTKeypoint]]  (outer = octave)
    SIFT.getDescriptors(keypointOctaves:)  SIFT.swift:207-238   -> [[SIFTDescriptor]]
    SIFTKeypoint / SIFTDescriptor          SIFT/SIFTKeypoint.swift:11-57, SIFT/SIFTDescriptor.swift:12-40
    IntegralSize                           Utilities/Math.swift:11-19
All compute happens in libsiftmi.so (hand-written HIP for gfx950, C ABI in include/siftmi.h); this
module only binds it.  There is no CPU fallback: importing works anywhere, creating a SIFT object
without a HIP device raises.
"""
import ctypes as C
import weakref
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from . import _capi
from ._capi import (FMT_BGRA8, FMT_GRAY8, FMT_GRAYF32, SiftmiError, descriptor_dtype, extremum_dtype,  # noqa: F401
                    keypoint_dtype, orientation_dtype)

__all__ = ["SIFT", "SIFTKeypoint", "SIFTDescriptor", "SIFTCorrespondence", "IntegralSize", "SiftmiError", "Engine"]


@dataclass(frozen=True)
class IntegralSize:                      # Utilities/Math.swift:11-19
    width: int
    height: int


@dataclass
class SIFTKeypoint:                      # SIFT/SIFTKeypoint.swift:11-57
    octave: int
    scale: int
    subScale: float
    scaledCoordinate: Tuple[int, int]
    absoluteCoordinate: Tuple[float, float]
    normalizedCoordinate: Tuple[float, float]
    sigma: float
    value: float


def _index_vectors(features):
    """rawFeatures, indexValue, indexKey of [n,128] integer features through siftmi_descriptor_index."""
    rec = np.zeros(len(features), descriptor_dtype)
    if len(features):
        rec["features"] = features
    raw, val, key = (np.zeros((len(rec), 128), np.float32), np.zeros((len(rec), 128), np.float32), np.zeros((len(rec), 16), np.float32))
    _capi.check(_capi.load().siftmi_descriptor_index(rec.ctypes.data, len(rec), raw.ctypes.data, val.ctypes.data, key.ctypes.data))
    return raw, val, key


@dataclass
class SIFTDescriptor:                    # SIFT/SIFTDescriptor.swift:12-89
    keypoint: SIFTKeypoint
    theta: float
    features: List[int]                  # 128 integers 0...255 (IntVector)
    rawFeatures: List[float] = field(default_factory=list, repr=False)   # features / 255 (FloatVector), :36-40
    indexValue: List[float] = field(default_factory=list, repr=False)    # cells re-ordered centre, corners, edges, :42-81 (private there)
    indexKey: List[float] = field(default_factory=list, repr=False)      # mean of each re-ordered cell, :83-87 (private there)

    def __post_init__(self):
        if len(self.features) == 0:      # precondition(features.count > 0), :31
            raise ValueError("features must not be empty")
        if len(self.rawFeatures) == 0 or len(self.indexValue) == 0 or len(self.indexKey) == 0:
            raw, val, key = _index_vectors(np.asarray(self.features, dtype=np.int64).reshape(1, 128))
            self.rawFeatures, self.indexValue, self.indexKey = raw[0], val[0], key[0]


@dataclass
class SIFTCorrespondence:                # SIFT/SIFTCorrespondence.swift:11-16
    source: SIFTDescriptor
    target: SIFTDescriptor
    featureDistance: float


def _fmt_of(img):
    if img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 4:
        return FMT_BGRA8
    if img.dtype == np.uint8 and img.ndim == 2:
        return FMT_GRAY8
    if img.dtype == np.float32 and img.ndim == 2:
        return FMT_GRAYF32
    raise ValueError("image must be HxWx4 uint8 (BGRA, the reference's .bgra8Unorm), HxW uint8 or HxW float32")


def _free_pinned(addr):
    try:
        _capi.load().siftmi_host_free(C.c_void_p(addr))
    except Exception:
        pass


def pinned_empty(shape, dtype=np.uint8):
    """numpy array over page-locked host memory (siftmi_host_alloc): frames stored here cross PCIe asynchronously at
    the full link rate when passed to the batch API.  The memory is freed when the array and every view of it are gone
    (the ctypes block that backs them carries a finalizer), or at once by pinned_release."""
    dtype = np.dtype(dtype)
    n = max(int(np.prod(shape)) * dtype.itemsize, 1)
    ptr = C.c_void_p()
    _capi.check(_capi.load().siftmi_host_alloc(n, C.byref(ptr)))
    block = (C.c_uint8 * n).from_address(ptr.value)      # numpy keeps `block` alive as the base of the array and its views
    block._siftmi_finalizer = weakref.finalize(block, _free_pinned, ptr.value)
    block._siftmi_finalizer.atexit = False               # at interpreter exit the pages are left to the process teardown (the HIP runtime may be gone)
    return np.frombuffer(block, dtype=np.uint8, count=int(np.prod(shape)) * dtype.itemsize).view(dtype).reshape(shape)


def pinned_release(arr):
    """Free a pinned_empty array's memory now (the array and its views must not be used afterwards)."""
    base = arr
    while getattr(base, "base", None) is not None:
        base = base.base
    fin = getattr(getattr(base, "obj", base), "_siftmi_finalizer", None)
    if fin is not None:
        fin()


class Engine:
    """Array-level access to one siftmi context (used by tests, bench.py and the stream driver)."""

    def __init__(self, width, height, device=0, **cfg):
        self.L = _capi.load()
        self.cfg = _capi.default_config(width, height, **cfg)
        h = C.c_void_p()
        _capi.check(self.L.siftmi_create(C.byref(self.cfg), device, C.byref(h)))
        self.h = h
        self.width, self.height = width, height
        self.n_octaves, self.nspo, self.max_batch = self.cfg.n_octaves, self.cfg.nspo, self.cfg.max_batch
        self.device = device

    def clone(self):
        """A second context with the same configuration on the same device (its own pyramid, lists and graphs).  Contexts
        are not re-entrant but independent ones run concurrently: stream.FrameStream(pipeline=2) alternates consecutive
        steps between an engine and its clone."""
        e = Engine.__new__(Engine)
        e.L = self.L
        e.cfg = type(self.cfg).from_buffer_copy(self.cfg)
        h = C.c_void_p()
        _capi.check(self.L.siftmi_create(C.byref(e.cfg), self.device, C.byref(h)))
        e.h = h
        e.width, e.height, e.device = self.width, self.height, self.device
        e.n_octaves, e.nspo, e.max_batch = self.n_octaves, self.nspo, self.max_batch
        return e

    def close(self):
        if getattr(self, "h", None):
            self.L.siftmi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- schedule ----
    def octave_size(self, o):
        w, h, d = C.c_int32(), C.c_int32(), C.c_float()
        _capi.check(self.L.siftmi_octave_size(self.h, o, C.byref(w), C.byref(h), C.byref(d)))
        return w.value, h.value, d.value

    def sigma(self, o, s):
        v = C.c_float()
        _capi.check(self.L.siftmi_get_sigma(self.h, o, s, C.byref(v)))
        return v.value

    def weights(self, layer):
        buf = np.zeros(32, np.float32)
        n = C.c_int32()
        _capi.check(self.L.siftmi_get_weights(self.h, layer, buf.ctypes.data, C.byref(n)))
        return buf[:n.value].copy()

    # ---- SIFT.getKeypoints / getDescriptors at array level ----
    def detect(self, img, allow_capacity=False):
        img = np.asarray(img)
        # rows may be strided (a view into a wider buffer: the C ABI takes a row stride); pixels of a row must be contiguous
        if not (img.strides[-1] == img.itemsize and (img.ndim == 2 or img.strides[1] == img.shape[2] * img.itemsize) and img.strides[0] > 0):
            img = np.ascontiguousarray(img)
        assert img.shape[0] == self.height and img.shape[1] == self.width, "image size != configured inputSize"
        out = C.c_void_p()
        counts = np.zeros(self.n_octaves, np.int32)
        _capi.check(self.L.siftmi_detect(self.h, img.ctypes.data, _fmt_of(img), img.strides[0], 0, C.byref(out),
                                         counts.ctypes.data_as(C.POINTER(C.c_int32))), allow_capacity)
        n = int(counts.sum())
        kps = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(n, 1) * 44,))[:n * 44].view(keypoint_dtype).copy()
        return kps, counts

    def describe(self, kps, counts, allow_capacity=False):
        kps = np.ascontiguousarray(kps, dtype=keypoint_dtype)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        out = C.c_void_p()
        dc = np.zeros(self.n_octaves, np.int32)
        _capi.check(self.L.siftmi_describe(self.h, kps.ctypes.data, counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(out),
                                           dc.ctypes.data_as(C.POINTER(C.c_int32))), allow_capacity)
        n = int(dc.sum())
        d = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(n, 1) * 136,))[:n * 136].view(descriptor_dtype).copy()
        return d, dc

    def detect_describe_batch(self, frames, allow_capacity=False, copy=True):
        """frames: [n, H, W(,4)] array.  Returns (keypoints, kp_counts[n, n_oct], descriptors, desc_counts[n, n_oct]).
        copy=False returns views of the context's pinned result buffers, valid until the next call (the C ABI's contract)."""
        frames = np.ascontiguousarray(frames)
        n = frames.shape[0]
        fmt = _fmt_of(frames[0])
        pk, pkc, pd, pdc = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _capi.check(self.L.siftmi_detect_describe_batch(self.h, n, frames.ctypes.data, fmt, frames.strides[1], frames.strides[0], 0,
                                                        C.byref(pk), C.byref(pkc), C.byref(pd), C.byref(pdc)), allow_capacity)
        kc = np.ctypeslib.as_array(C.cast(pkc, C.POINTER(C.c_int32)), shape=(n, self.n_octaves)).copy()
        dc = np.ctypeslib.as_array(C.cast(pdc, C.POINTER(C.c_int32)), shape=(n, self.n_octaves)).copy()
        nk, nd = int(kc.sum()), int(dc.sum())
        kps = np.ctypeslib.as_array(C.cast(pk, C.POINTER(C.c_uint8)), shape=(max(nk, 1) * 44,))[:nk * 44].view(keypoint_dtype)
        ds = np.ctypeslib.as_array(C.cast(pd, C.POINTER(C.c_uint8)), shape=(max(nd, 1) * 136,))[:nd * 136].view(descriptor_dtype)
        if copy:
            kps, ds = kps.copy(), ds.copy()
        return kps, kc, ds, dc

    def detect_describe_batch_device(self, n_frames, d_pixels, fmt, row_stride, frame_stride, d_kp, kp_cap, d_desc, desc_cap,
                                     d_counts, d_totals, stream=None):
        """All pointers are device addresses (ints); asynchronous on `stream`."""
        _capi.check(self.L.siftmi_detect_describe_batch_device(self.h, n_frames, d_pixels, fmt, row_stride, frame_stride, d_kp, kp_cap,
                                                               d_desc, desc_cap, d_counts, d_totals, stream))

    # ---- SIFTDescriptor.match (next row) ----
    def match(self, source, target, absolute_threshold=1.176, relative_threshold=0.6):
        """source / target: siftmi descriptor records (descriptor_dtype).  Returns match records in source order."""
        a = np.ascontiguousarray(source, dtype=descriptor_dtype)
        b = np.ascontiguousarray(target, dtype=descriptor_dtype)
        out, n = C.c_void_p(), C.c_int64()
        _capi.check(self.L.siftmi_match_descriptors(self.h, a.ctypes.data, len(a), b.ctypes.data, len(b), 0, absolute_threshold,
                                                    relative_threshold, C.byref(out), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, _capi.match_dtype)
        return np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(n.value * 12,)).view(_capi.match_dtype).copy()

    def match_device(self, d_source, n_source, d_target, n_target, d_matches, d_count, absolute_threshold=1.176, relative_threshold=0.6, stream=None):
        """siftmi_match_descriptors_device: all pointers are device addresses (ints); asynchronous, nothing is synchronised."""
        _capi.check(self.L.siftmi_match_descriptors_device(self.h, d_source, n_source, d_target, n_target, absolute_threshold, relative_threshold,
                                                          d_matches, d_count, stream))

    def match_plan(self, n_source, n_target):
        """(targets per chunk, chunks, chunks start from a bound) for a problem of this size (siftmi_match_plan)."""
        sl, ns, b = C.c_int64(), C.c_int64(), C.c_int()
        _capi.check(self.L.siftmi_match_plan(n_source, n_target, C.byref(sl), C.byref(ns), C.byref(b)))
        return sl.value, ns.value, bool(b.value)

    def approximate_match(self, source, target, absolute_threshold=300.0, relative_threshold=0.6):
        """SIFTDescriptor.approximateMatch on descriptor records -> match records in source order."""
        a = np.ascontiguousarray(source, dtype=descriptor_dtype)
        b = np.ascontiguousarray(target, dtype=descriptor_dtype)
        out, n = C.c_void_p(), C.c_int64()
        _capi.check(self.L.siftmi_approximate_match(self.h, a.ctypes.data, len(a), b.ctypes.data, len(b), 0, absolute_threshold,
                                                    relative_threshold, C.byref(out), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, _capi.match_dtype)
        return np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(n.value * 12,)).view(_capi.match_dtype).copy()

    def match_geometry(self, source, source_xy, target, target_xy, absolute_threshold=1.176, relative_threshold=0.6):
        """SIFTDescriptor.matchGeometry on descriptor records + [n,2] (x, y) absolute coordinates -> (score, n_matches)."""
        a = np.ascontiguousarray(source, dtype=descriptor_dtype)
        b = np.ascontiguousarray(target, dtype=descriptor_dtype)
        axy = np.ascontiguousarray(source_xy, dtype=np.float32)
        bxy = np.ascontiguousarray(target_xy, dtype=np.float32)
        assert axy.shape == (len(a), 2) and bxy.shape == (len(b), 2)
        score, n = C.c_float(), C.c_int64()
        _capi.check(self.L.siftmi_match_geometry(self.h, a.ctypes.data, axy.ctypes.data, len(a), b.ctypes.data, bxy.ctypes.data, len(b),
                                                 absolute_threshold, relative_threshold, C.byref(score), C.byref(n)))
        return float(score.value), int(n.value)

    # ---- introspection ----
    def stats(self):
        s = _capi.Stats()
        _capi.check(self.L.siftmi_get_stats(self.h, C.byref(s)))
        shape = (s.n_frames, s.n_octaves)
        out = {k: np.ctypeslib.as_array(getattr(s, k), shape=shape).copy()
               for k in ("raw_extrema", "candidates", "keypoints", "oriented", "descriptors")}
        out["raw_extrema_exact"] = bool(s.raw_extrema_exact)
        return out

    def graph_stats(self):
        """Launch sequences of the batched entry points since the context was created: captured / replayed / issued directly."""
        a, b, c, f = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        _capi.check(self.L.siftmi_graph_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(f)))
        return {"captures": int(a.value), "replays": int(b.value), "direct": int(c.value), "last_replayed": bool(f.value & 1),
                "last_forked": bool(f.value & 2), "dense_hint": bool(f.value & 4)}

    def gaussian(self, o, s, frame=0):
        w, h, _ = self.octave_size(o)
        out = np.empty((h, w), np.float32)
        _capi.check(self.L.siftmi_copy_gaussian(self.h, frame, o, s, out.ctypes.data))
        return out

    def dog(self, o, s, frame=0):
        """DoG layer D[o][s] = G[s + 1] - G[s] (the reference's DifferenceOfGaussians textures)."""
        w, h, _ = self.octave_size(o)
        out = np.empty((h, w), np.float32)
        _capi.check(self.L.siftmi_copy_dog(self.h, frame, o, s, out.ctypes.data))
        return out

    def extrema(self, o, frame=0):
        n = C.c_int32()
        _capi.check(self.L.siftmi_copy_extrema(self.h, frame, o, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), extremum_dtype)
        _capi.check(self.L.siftmi_copy_extrema(self.h, frame, o, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def orientations(self, o, frame=0):
        n = C.c_int32()
        _capi.check(self.L.siftmi_copy_orientations(self.h, frame, o, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), orientation_dtype)
        _capi.check(self.L.siftmi_copy_orientations(self.h, frame, o, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def descriptor_floats(self, o, frame=0):
        n = C.c_int32()
        _capi.check(self.L.siftmi_copy_descriptor_floats(self.h, frame, o, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 128), np.float32)
        _capi.check(self.L.siftmi_copy_descriptor_floats(self.h, frame, o, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    # ---- timing ----
    def enable_timings(self, on=True):
        _capi.check(self.L.siftmi_enable_timings(self.h, int(on)))

    def reset_timings(self):
        _capi.check(self.L.siftmi_reset_timings(self.h))

    def timings(self):
        ms = np.zeros(len(_capi.T_NAMES), np.float64)
        n = np.zeros(len(_capi.T_NAMES), np.int64)
        _capi.check(self.L.siftmi_get_timings(self.h, ms.ctypes.data, n.ctypes.data))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(_capi.T_NAMES)}

    def blur_layer_timings(self, o, layer):
        """(accumulated ms, launches, uses the marching kernel) of the blur launches of (octave, layer) since reset_timings."""
        ms, n, m = C.c_double(), C.c_int64(), C.c_int32()
        _capi.check(self.L.siftmi_get_blur_layer_timings(self.h, o, layer, C.byref(ms), C.byref(n), C.byref(m)))
        return ms.value, n.value, bool(m.value & 1)

    def blur_layer_kind(self, o, layer):
        """Which kernel the last call used for (octave, layer) and whether it wrote the extrema scan's activity flags."""
        ms, n, m = C.c_double(), C.c_int64(), C.c_int32()
        _capi.check(self.L.siftmi_get_blur_layer_timings(self.h, o, layer, C.byref(ms), C.byref(n), C.byref(m)))
        return {"kernel": "blur_chain_kernel" if m.value & 4 else ("blur_ring_kernel" if m.value & 1 else "blur2_kernel"), "activity_flags": bool(m.value & 2)}

    def blur_algorithmic_bytes(self, o):
        return int(self.L.siftmi_blur_algorithmic_bytes(self.h, o))

    def time_blur(self, o, layer, iters=20):
        v = C.c_double()
        _capi.check(self.L.siftmi_time_blur(self.h, o, layer, iters, C.byref(v)))
        return v.value

    def time_copy(self, nbytes, iters=10):
        """(ms per launch, bytes moved per launch) of a plain float4 copy inside the pyramid memory: the measured HBM ceiling."""
        v, n = C.c_double(), C.c_int64()
        _capi.check(self.L.siftmi_time_copy(self.h, int(nbytes), iters, C.byref(v), C.byref(n)))
        return v.value, int(n.value)

    def time_blur_memory(self, o, layer, iters=10):
        """ms per launch of the layer's ring kernel with its arithmetic compiled out (loads, LDS staging, barriers, stores only)."""
        v = C.c_double()
        _capi.check(self.L.siftmi_time_blur_memory(self.h, o, layer, iters, C.byref(v)))
        return v.value

    def synchronize(self):
        _capi.check(self.L.siftmi_synchronize(self.h))


def _kp_obj(r):
    return SIFTKeypoint(int(r["octave"]), int(r["scale"]), float(r["sub_scale"]), (int(r["x"]), int(r["y"])),
                        (float(r["abs_x"]), float(r["abs_y"])), (float(r["norm_x"]), float(r["norm_y"])),
                        float(r["sigma"]), float(r["value"]))


def _kp_rec(k):
    return (k.octave, k.scale, k.subScale, k.scaledCoordinate[0], k.scaledCoordinate[1], k.absoluteCoordinate[0],
            k.absoluteCoordinate[1], k.normalizedCoordinate[0], k.normalizedCoordinate[1], k.sigma, k.value)


class SIFT:
    """Mirror of `public final class SIFT` (Sources/SIFTMetal/SIFT/SIFT.swift:55)."""

    @dataclass
    class Configuration:                 # SIFT.swift:57-103 (+ the octave count the reference hard-wires to 7)
        inputSize: IntegralSize
        numberOfOctaves: int = 7
        numberOfScalesPerOctave: int = 3

    def __init__(self, device=0, configuration=None):
        """device: HIP device ordinal (the reference takes an MTLDevice)."""
        if configuration is None:
            raise ValueError("configuration is required (SIFT.Configuration(inputSize: IntegralSize))")
        self.configuration = configuration
        self._engine = Engine(configuration.inputSize.width, configuration.inputSize.height, device=device,
                              n_octaves=configuration.numberOfOctaves, nspo=configuration.numberOfScalesPerOctave)

    def getKeypoints(self, inputTexture) -> List[List[SIFTKeypoint]]:
        """inputTexture: HxWx4 uint8 BGRA array (the reference requires a .bgra8Unorm MTLTexture of
        exactly inputSize); HxW uint8 / float32 luma arrays are accepted as an extension."""
        kps, counts = self._engine.detect(inputTexture)
        out, pos = [], 0
        for o in range(self._engine.n_octaves):
            out.append([_kp_obj(r) for r in kps[pos:pos + counts[o]]])
            pos += counts[o]
        return out

    def getDescriptors(self, keypointOctaves: List[List[SIFTKeypoint]]) -> List[List[SIFTDescriptor]]:
        if len(keypointOctaves) != self._engine.n_octaves:      # precondition, SIFT.swift:208
            raise ValueError("keypointOctaves.count must equal the number of octaves")
        counts = np.array([len(kk) for kk in keypointOctaves], np.int32)
        flat = np.array([_kp_rec(k) for kk in keypointOctaves for k in kk], dtype=keypoint_dtype)
        ds, dc = self._engine.describe(flat, counts)
        out, pos = [], 0
        for o in range(self._engine.n_octaves):
            rows = ds[pos:pos + dc[o]]
            raw, val, key = _index_vectors(rows["features"])
            out.append([SIFTDescriptor(keypointOctaves[o][int(r["keypoint"])], float(r["theta"]), r["features"].astype(int).tolist(),
                                       raw[i], val[i], key[i]) for i, r in enumerate(rows)])
            pos += dc[o]
        return out

    @staticmethod
    def _pack(ds):
        out = np.zeros(len(ds), descriptor_dtype)
        for i, d in enumerate(ds):
            out[i]["theta"] = d.theta
            out[i]["features"] = d.features
        return out

    def match(self, source: List[SIFTDescriptor], target: List[SIFTDescriptor], absoluteThreshold: float = 1.176,
              relativeThreshold: float = 0.6) -> List["SIFTCorrespondence"]:
        """SIFTDescriptor.match(source:target:absoluteThreshold:relativeThreshold:) (SIFTDescriptor.swift:298-318);
        a static function in the reference, hosted on the SIFT object here because it needs the device context."""
        m = self._engine.match(self._pack(source), self._pack(target), absoluteThreshold, relativeThreshold)
        return [SIFTCorrespondence(source[int(r["source"])], target[int(r["target"])], float(r["distance"])) for r in m]

    def approximateMatch(self, source: List[SIFTDescriptor], target: List[SIFTDescriptor], absoluteThreshold: float = 300,
                         relativeThreshold: float = 0.6) -> List["SIFTCorrespondence"]:
        """SIFTDescriptor.approximateMatch(source:target:absoluteThreshold:relativeThreshold:) (SIFTDescriptor.swift:362-388)."""
        m = self._engine.approximate_match(self._pack(source), self._pack(target), absoluteThreshold, relativeThreshold)
        return [SIFTCorrespondence(source[int(r["source"])], target[int(r["target"])], float(r["distance"])) for r in m]

    def matchGeometry(self, source: List[SIFTDescriptor], target: List[SIFTDescriptor], absoluteThreshold: float = 1.176,
                      relativeThreshold: float = 0.6) -> float:
        """SIFTDescriptor.matchGeometry(source:target:absoluteThreshold:relativeThreshold:) (SIFTDescriptor.swift:104-144)."""
        a, b = self._pack(source), self._pack(target)
        axy = np.array([d.keypoint.absoluteCoordinate for d in source], np.float32).reshape(-1, 2)
        bxy = np.array([d.keypoint.absoluteCoordinate for d in target], np.float32).reshape(-1, 2)
        return self._engine.match_geometry(a, axy, b, bxy, absoluteThreshold, relativeThreshold)[0]

    # BASELINE.json's north_star names the API detect()/describe(); keep them as aliases.
    detect = getKeypoints
    describe = getDescriptors
