"""ctypes loader for the CPU restatement (oracle/sift_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (siftmetal_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("SIFT_ORACLE_LIB") or os.path.join(_HERE, "libsift_oracle.so")     # SIFT_ORACLE_LIB: the sanitizer build (make asan)

FMT_BGRA8, FMT_GRAY8, FMT_GRAYF32 = 0, 1, 2


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("n_octaves", C.c_int32),
                ("nspo", C.c_int32), ("full_neighbourhood", C.c_int32), ("use_fma", C.c_int32)]


extremum_dtype = np.dtype([("x", "<i4"), ("y", "<i4"), ("scale", "<i4")])
keypoint_dtype = np.dtype([("octave", "<i4"), ("scale", "<i4"), ("subScale", "<f4"),
                           ("x", "<i4"), ("y", "<i4"), ("absX", "<f4"), ("absY", "<f4"),
                           ("normX", "<f4"), ("normY", "<f4"), ("sigma", "<f4"), ("value", "<f4")])
orientation_dtype = np.dtype([("keypoint", "<i4"), ("count", "<i4"), ("orientations", "<f4", (36,))])
descriptor_dtype = np.dtype([("valid", "<i4"), ("keypoint", "<i4"), ("theta", "<f4"),
                             ("features", "<i4", (128,))])
assert keypoint_dtype.itemsize == 44 and orientation_dtype.itemsize == 152 and descriptor_dtype.itemsize == 524


def build(force=False):
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "sift_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if _LIB.endswith("_asan.so") else []))
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.so_create.restype = C.c_void_p
        L.so_create.argtypes = [C.POINTER(Config)]
        L.so_destroy.argtypes = [C.c_void_p]
        for f in ("so_octave_width", "so_octave_height"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        L.so_octave_delta.restype = C.c_float
        L.so_octave_delta.argtypes = [C.c_void_p, C.c_int]
        L.so_octave_sigma.restype = C.c_float
        L.so_octave_sigma.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.so_blur_taps.restype = C.c_int
        L.so_blur_taps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.so_build_pyramid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        for f in ("so_gaussian", "so_dog"):
            getattr(L, f).restype = C.POINTER(C.c_float)
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.so_seed.restype = C.POINTER(C.c_float)
        L.so_seed.argtypes = [C.c_void_p]
        L.so_extrema.restype = C.c_int
        L.so_extrema.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.so_refine.restype = C.c_int
        L.so_refine.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.so_refine_stages.restype = C.c_int
        L.so_refine_stages.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.so_orientations.restype = C.c_int
        L.so_orientations.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.so_orientation_histogram.restype = None
        L.so_orientation_histogram.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.so_descriptors.restype = C.c_int
        L.so_descriptors.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int]
        L.so_detect_describe.restype = C.c_int
        L.so_detect_describe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4
        L.so_num_threads.restype = C.c_int
        L.so_match.restype = C.c_int
        L.so_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int]
        L.so_descriptor_index.restype = None
        L.so_descriptor_index.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.so_compare_geometry.restype = C.c_float
        L.so_compare_geometry.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.so_match_geometry.restype = C.c_float
        L.so_match_geometry.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                        C.POINTER(C.c_int)]
        L.so_trie_create.restype = C.c_void_p
        L.so_trie_create.argtypes = [C.c_int, C.c_void_p]
        L.so_trie_destroy.argtypes = [C.c_void_p]
        L.so_trie_insert.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.so_trie_contains.restype = C.c_int
        L.so_trie_contains.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.so_trie_capacity.restype = C.c_int
        L.so_trie_capacity.argtypes = [C.c_void_p]
        L.so_trie_link.restype = C.c_int
        L.so_trie_link.argtypes = [C.c_void_p]
        L.so_trie_nearest.restype = C.c_int
        L.so_trie_nearest.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.so_approximate_match.restype = C.c_int
        L.so_approximate_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _fmt_of(img):
    if img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 4:
        return FMT_BGRA8
    if img.dtype == np.uint8 and img.ndim == 2:
        return FMT_GRAY8
    if img.dtype == np.float32 and img.ndim == 2:
        return FMT_GRAYF32
    raise ValueError("image must be HxWx4 u8 (BGRA), HxW u8 or HxW f32")


class Oracle:
    """Per-stage access to the restatement (mirrors the reference's stage boundaries)."""

    def __init__(self, width, height, n_octaves=7, nspo=3, full_neighbourhood=False, use_fma=True):
        self.cfg = Config(width, height, n_octaves, nspo, int(full_neighbourhood), int(use_fma))
        self.L = lib()
        self.h = self.L.so_create(C.byref(self.cfg))
        if not self.h:
            raise ValueError("so_create failed")
        self.n_octaves, self.nspo = n_octaves, nspo

    def close(self):
        if self.h:
            self.L.so_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def octave_size(self, o):
        return self.L.so_octave_width(self.h, o), self.L.so_octave_height(self.h, o)

    def delta(self, o):
        return self.L.so_octave_delta(self.h, o)

    def sigma(self, o, s):
        return self.L.so_octave_sigma(self.h, o, s)

    def weights(self, layer):
        buf = np.zeros(32, np.float32)
        n = self.L.so_blur_taps(self.h, layer, _ptr(buf))
        return buf[:n].copy()

    def build_pyramid(self, img):
        img = np.ascontiguousarray(img)
        assert img.shape[0] == self.cfg.height and img.shape[1] == self.cfg.width
        self.L.so_build_pyramid(self.h, _ptr(img), _fmt_of(img), img.strides[0])

    def _view(self, p, o):
        w, h = self.octave_size(o)
        return np.ctypeslib.as_array(p, shape=(h, w))

    def gaussian(self, o, s):
        return self._view(self.L.so_gaussian(self.h, o, s), o)

    def dog(self, o, s):
        return self._view(self.L.so_dog(self.h, o, s), o)

    def seed(self):
        return self._view(self.L.so_seed(self.h), 0)

    def extrema(self, o):
        n = self.L.so_extrema(self.h, o, None, 0)
        out = np.zeros(n, extremum_dtype)
        self.L.so_extrema(self.h, o, _ptr(out), n)
        return out

    def refine(self, o, ext):
        ext = np.ascontiguousarray(ext)
        out = np.zeros(len(ext), keypoint_dtype)
        n = self.L.so_refine(self.h, o, _ptr(ext), len(ext), _ptr(out), len(out))
        return out[:n].copy()

    def refine_stages(self, o, ext, contrast_terms=1):
        """(keypoints, [pre-filter, converged, contrast, edge] survivor counts, rows [n, 4] = y, x, sigma, stage reached per
        candidate); contrast_terms = 3: IPOL's three-term contrast instead of the reference's x-term-only one."""
        ext = np.ascontiguousarray(ext)
        out = np.zeros(len(ext), keypoint_dtype)
        st = np.zeros(4, np.int32)
        rows = np.zeros((len(ext), 4), np.float32)
        n = self.L.so_refine_stages(self.h, o, _ptr(ext), len(ext), _ptr(out), len(out), _ptr(st), int(contrast_terms), _ptr(rows))
        return out[:n].copy(), st.tolist(), rows

    def orientations(self, o, kp):
        kp = np.ascontiguousarray(kp)
        out = np.zeros(len(kp), orientation_dtype)
        n = self.L.so_orientations(self.h, o, _ptr(kp), len(kp), _ptr(out), len(out))
        return out[:n].copy()

    def orientation_histogram(self, o, kp_record):
        k = np.ascontiguousarray(np.array([kp_record], dtype=keypoint_dtype))
        h = np.zeros(36, np.float32)
        self.L.so_orientation_histogram(self.h, o, _ptr(k), _ptr(h))
        return h

    def descriptors(self, o, kp, ori, want_float=False):
        kp = np.ascontiguousarray(kp)
        ori = np.ascontiguousarray(ori)
        nd = int(ori["count"].sum()) if len(ori) else 0
        out = np.zeros(nd, descriptor_dtype)
        f32 = np.zeros((nd, 128), np.float32) if want_float else None
        n = self.L.so_descriptors(self.h, o, _ptr(kp), _ptr(ori), len(ori), _ptr(out), _ptr(f32), nd)
        assert n == nd
        return (out, f32) if want_float else out

    def detect_describe_counts(self, img):
        img = np.ascontiguousarray(img)
        c = [np.zeros(self.n_octaves, np.int32) for _ in range(4)]
        total = self.L.so_detect_describe(self.h, _ptr(img), _fmt_of(img), img.strides[0], *[_ptr(a) for a in c])
        return total, c

    def run(self, img, want_float=False):
        """Full path; returns per-octave dicts of extrema / keypoints / orientations / descriptors."""
        self.build_pyramid(img)
        res = []
        for o in range(self.n_octaves):
            ext = self.extrema(o)
            kp = self.refine(o, ext)
            ori = self.orientations(o, kp)
            d = self.descriptors(o, kp, ori, want_float)
            res.append({"extrema": ext, "keypoints": kp, "orientations": ori,
                        "descriptors": d[0] if want_float else d,
                        "features_f32": d[1] if want_float else None})
        return res


match_dtype = np.dtype([("source", "<i4"), ("target", "<i4"), ("distance", "<f4")])


def match(src_features, tgt_features, absolute_threshold=1.176, relative_threshold=0.6):
    """SIFTDescriptor.match on [n,128] integer feature arrays -> structured array of correspondences."""
    a = np.ascontiguousarray(src_features, dtype=np.int32)
    b = np.ascontiguousarray(tgt_features, dtype=np.int32)
    out = np.zeros(max(len(a), 1), match_dtype)
    n = lib().so_match(_ptr(a), len(a), _ptr(b), len(b), absolute_threshold, relative_threshold, _ptr(out), len(out))
    return out[:n].copy()


def descriptor_index(features):
    """SIFTDescriptor.init's derived vectors: (rawFeatures [n,128], indexValue [n,128], indexKey [n,16])."""
    f = np.ascontiguousarray(features, dtype=np.int32).reshape(-1, 128)
    raw, val, key = np.zeros((len(f), 128), np.float32), np.zeros((len(f), 128), np.float32), np.zeros((len(f), 16), np.float32)
    lib().so_descriptor_index(_ptr(f), len(f), _ptr(raw), _ptr(val), _ptr(key))
    return raw, val, key


def compare_geometry(matches, src_xy, tgt_xy, minimum_sample_size=7):
    m = np.ascontiguousarray(matches, dtype=match_dtype)
    a, b = np.ascontiguousarray(src_xy, dtype=np.float32), np.ascontiguousarray(tgt_xy, dtype=np.float32)
    return float(lib().so_compare_geometry(_ptr(m), len(m), _ptr(a), _ptr(b), minimum_sample_size))


def match_geometry(src_features, src_xy, tgt_features, tgt_xy, absolute_threshold=1.176, relative_threshold=0.6):
    """SIFTDescriptor.matchGeometry -> (score, number of matches)."""
    a = np.ascontiguousarray(src_features, dtype=np.int32)
    b = np.ascontiguousarray(tgt_features, dtype=np.int32)
    axy, bxy = np.ascontiguousarray(src_xy, dtype=np.float32), np.ascontiguousarray(tgt_xy, dtype=np.float32)
    n = C.c_int(0)
    s = lib().so_match_geometry(_ptr(a), _ptr(axy), len(a), _ptr(b), _ptr(bxy), len(b), absolute_threshold, relative_threshold, C.byref(n))
    return float(s), n.value


class Trie:
    """Utilities/Trie.swift with descriptor ids as values (features: [n,128] ints the ids index)."""

    def __init__(self, number_of_bins, features=None):
        self.features = np.ascontiguousarray(features if features is not None else np.zeros((1, 128)), dtype=np.int32)
        self.h = lib().so_trie_create(number_of_bins, _ptr(self.features))

    def __del__(self):
        try:
            lib().so_trie_destroy(self.h)
        except Exception:
            pass

    def insert(self, key, value):
        k = np.ascontiguousarray(key, dtype=np.float32)
        lib().so_trie_insert(self.h, _ptr(k), len(k), int(value))

    def contains(self, key):
        k = np.ascontiguousarray(key, dtype=np.float32)
        return bool(lib().so_trie_contains(self.h, _ptr(k), len(k)))

    def capacity(self):
        return lib().so_trie_capacity(self.h)

    def link(self):
        return lib().so_trie_link(self.h)

    def nearest(self, key, query, radius, k):
        kk = np.ascontiguousarray(key, dtype=np.float32)
        q = np.ascontiguousarray(query, dtype=np.int32)
        ids, dist = np.zeros(8, np.int32), np.zeros(8, np.float32)
        n = lib().so_trie_nearest(self.h, _ptr(kk), len(kk), _ptr(q), radius, k, _ptr(ids), _ptr(dist))
        return list(zip(ids[:n].tolist(), dist[:n].tolist()))


def approximate_match(src_features, tgt_features, absolute_threshold=300.0, relative_threshold=0.6):
    """SIFTDescriptor.approximateMatch on [n,128] integer feature arrays -> structured array of correspondences."""
    a = np.ascontiguousarray(src_features, dtype=np.int32)
    b = np.ascontiguousarray(tgt_features, dtype=np.int32)
    out = np.zeros(max(len(a), 1), match_dtype)
    n = lib().so_approximate_match(_ptr(a), len(a), _ptr(b), len(b), absolute_threshold, relative_threshold, _ptr(out), len(out))
    return out[:n].copy()


def num_threads():
    return lib().so_num_threads()


def set_num_threads(n):
    lib().so_set_num_threads(int(n))
