"""ctypes binding of libsiftmi.so (include/siftmi.h).  No fallback: if the library is missing or
no HIP device is visible, this raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SIFTMI_LIB") or os.path.join(_HERE, "libsiftmi.so")      # SIFTMI_LIB: an experiment build (tools/)

OK, E_BADARG, E_CAPACITY, E_HIP, E_NODEVICE, E_NOMEM, E_STATE = 0, -1, -2, -3, -4, -5, -6
FMT_BGRA8, FMT_GRAY8, FMT_GRAYF32 = 0, 1, 2
T_NAMES = ["seed", "blur", "downsample", "extrema", "refine", "sort", "orient", "describe", "pack"]

EXPORTS = [
    "siftmi_default_config", "siftmi_create", "siftmi_destroy", "siftmi_last_error", "siftmi_device_count",
    "siftmi_detect", "siftmi_describe", "siftmi_detect_describe_batch", "siftmi_detect_describe_batch_device",
    "siftmi_descriptor_to_reference", "siftmi_host_alloc", "siftmi_host_free", "siftmi_match_descriptors", "siftmi_match_descriptors_device", "siftmi_match_plan", "siftmi_approximate_match", "siftmi_match_geometry", "siftmi_descriptor_index", "siftmi_get_stats", "siftmi_octave_size", "siftmi_get_sigma",
    "siftmi_get_weights", "siftmi_copy_gaussian", "siftmi_copy_dog", "siftmi_copy_extrema", "siftmi_copy_orientations",
    "siftmi_copy_descriptor_floats", "siftmi_enable_timings", "siftmi_reset_timings", "siftmi_get_timings",
    "siftmi_blur_algorithmic_bytes", "siftmi_get_blur_layer_timings", "siftmi_time_blur", "siftmi_time_copy", "siftmi_time_blur_memory", "siftmi_synchronize",
    "siftmi_device_alloc", "siftmi_device_free", "siftmi_memcpy", "siftmi_device_synchronize",
    "siftmi_stream_default_config", "siftmi_stream_create", "siftmi_stream_destroy", "siftmi_stream_context",
    "siftmi_stream_submit_device", "siftmi_stream_submit_host", "siftmi_stream_wait_upload", "siftmi_stream_wait_consumed",
    "siftmi_stream_result_device", "siftmi_stream_result_host", "siftmi_stream_synchronize", "siftmi_stream_set_density_mode",
    "siftmi_exchange_unique_id", "siftmi_exchange_create", "siftmi_exchange_destroy", "siftmi_exchange_gather",
    "siftmi_exchange_result", "siftmi_exchange_finish", "siftmi_exchange_stats", "siftmi_exchange_set_headroom", "siftmi_exchange_transport",
    "siftmi_exchange_ranks", "siftmi_exchange_set_timeout", "siftmi_exchange_wait", "siftmi_graph_stats",
    "siftmi_gather_plan_init", "siftmi_gather_plan_resolve",
]
NO_STREAM = C.c_void_p(-1).value          # SIFTMI_NO_STREAM
UNIQUE_ID_BYTES = 128


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("n_octaves", C.c_int32), ("nspo", C.c_int32),
                ("sigma_min", C.c_float), ("delta_min", C.c_float), ("sigma_in", C.c_float),
                ("dog_threshold", C.c_float), ("edge_threshold", C.c_float), ("max_iterations", C.c_int32),
                ("max_offset", C.c_float), ("image_border", C.c_int32), ("lambda_orientation", C.c_float),
                ("orientation_threshold", C.c_float), ("orientation_smoothing", C.c_int32),
                ("descriptor_scales_per_octave", C.c_int32), ("full_neighbourhood", C.c_int32),
                ("max_batch", C.c_int32), ("max_extrema", C.c_int32), ("max_keypoints", C.c_int32),
                ("max_descriptors", C.c_int32), ("keep_descriptor_floats", C.c_int32), ("use_hip_graph", C.c_int32), ("count_raw_extrema", C.c_int32),
                ("blur_march_min_blocks", C.c_int32), ("blur_chain_max_tiles", C.c_int32), ("graph_fork", C.c_int32), ("descriptor_patch_lds", C.c_int32), ("reserved", C.c_int32 * 1)]


class Stats(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_octaves", C.c_int32)] + \
               [(n, C.POINTER(C.c_int32)) for n in ("raw_extrema", "candidates", "keypoints", "oriented", "descriptors")] + \
               [("raw_extrema_exact", C.c_int32)]


class StreamConfig(C.Structure):         # siftmi_stream_config
    _fields_ = [("frames_per_step", C.c_int32), ("steps_in_flight", C.c_int32), ("result_sets", C.c_int32), ("format", C.c_int32),
                ("kp_per_frame", C.c_int64), ("desc_per_frame", C.c_int64), ("staging_buffers", C.c_int32), ("density_mode", C.c_int32),
                ("reserved", C.c_int32 * 6)]


class StepDevice(C.Structure):           # siftmi_step_device
    _fields_ = [("step", C.c_int64), ("keypoints", C.c_void_p), ("descriptors", C.c_void_p), ("counts", C.c_void_p), ("totals", C.c_void_p),
                ("kp_capacity", C.c_int64), ("desc_capacity", C.c_int64)]


class StepHost(C.Structure):             # siftmi_step_host
    _fields_ = [("step", C.c_int64), ("keypoints", C.c_void_p), ("descriptors", C.c_void_p), ("counts", C.POINTER(C.c_int32)),
                ("n_keypoints", C.c_int32), ("n_descriptors", C.c_int32), ("overflow_flags", C.c_int32), ("launch_flags", C.c_int32)]


STEP_DENSE_HINT, STEP_GRAPH_REPLAY, STEP_FORKED, STEP_RAW_EXACT = 1, 2, 4, 8      # siftmi_step_host.launch_flags


class Gathered(C.Structure):             # siftmi_gathered
    _fields_ = [("step", C.c_int64), ("world", C.c_int32), ("complete", C.c_int32), ("keypoints", C.c_void_p), ("descriptors", C.c_void_p),
                ("counts", C.c_void_p), ("totals_device", C.c_void_p), ("totals_host", C.POINTER(C.c_int32)),
                ("kp_stride", C.c_int64), ("desc_stride", C.c_int64), ("kp_records", C.c_int64), ("desc_records", C.c_int64),
                ("resolved", C.c_int32), ("reserved", C.c_int32)]


class GatherPlan(C.Structure):           # siftmi_gather_plan
    _fields_ = [("kp_capacity", C.c_int64), ("desc_capacity", C.c_int64), ("send_kp", C.c_int64), ("send_desc", C.c_int64),
                ("quantum", C.c_int64), ("headroom_percent", C.c_int32), ("reserved", C.c_int32),
                ("steps_resolved", C.c_int64), ("steps_incomplete", C.c_int64), ("steps_overflowed", C.c_int64)]


extremum_dtype = np.dtype([("x", "<i4"), ("y", "<i4"), ("scale", "<i4")])
keypoint_dtype = np.dtype([("octave", "<i4"), ("scale", "<i4"), ("sub_scale", "<f4"), ("x", "<i4"), ("y", "<i4"),
                           ("abs_x", "<f4"), ("abs_y", "<f4"), ("norm_x", "<f4"), ("norm_y", "<f4"),
                           ("sigma", "<f4"), ("value", "<f4")])
orientation_dtype = np.dtype([("keypoint", "<i4"), ("count", "<i4"), ("orientations", "<f4", (36,))])
descriptor_dtype = np.dtype([("keypoint", "<i4"), ("theta", "<f4"), ("features", "u1", (128,))])
match_dtype = np.dtype([("source", "<i4"), ("target", "<i4"), ("distance", "<f4")])
descriptor_reference_dtype = np.dtype([("valid", "<i4"), ("keypoint", "<i4"), ("theta", "<f4"), ("features", "<i4", (128,))])
assert keypoint_dtype.itemsize == 44 and descriptor_dtype.itemsize == 136
assert orientation_dtype.itemsize == 152 and descriptor_reference_dtype.itemsize == 524


class SiftmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("siftmi error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load libsiftmi.so (built in-tree by __graft_entry__.build() / make -C siftmetal_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libsiftmi.so not built: run `make -C siftmetal_amd/csrc` (hipcc, gfx950). "
                          "There is no CPU fallback for the product path.")
    L = C.CDLL(LIB_PATH)
    vp, i32p = C.c_void_p, C.POINTER(C.c_int32)
    L.siftmi_default_config.argtypes = [C.POINTER(Config), C.c_int32, C.c_int32]
    L.siftmi_create.argtypes = [C.POINTER(Config), C.c_int, C.POINTER(vp)]
    L.siftmi_destroy.argtypes = [vp]
    L.siftmi_destroy.restype = None
    L.siftmi_last_error.restype = C.c_char_p
    L.siftmi_detect.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_int, C.POINTER(vp), i32p]
    L.siftmi_describe.argtypes = [vp, vp, i32p, C.POINTER(vp), i32p]
    L.siftmi_detect_describe_batch.argtypes = [vp, C.c_int32, vp, C.c_int, C.c_size_t, C.c_size_t, C.c_int,
                                               C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.siftmi_detect_describe_batch_device.argtypes = [vp, C.c_int32, vp, C.c_int, C.c_size_t, C.c_size_t,
                                                      vp, C.c_int64, vp, C.c_int64, vp, vp, vp]
    L.siftmi_match_descriptors.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_float, C.c_float, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.siftmi_match_descriptors_device.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.c_float, C.c_float, vp, vp, vp]
    L.siftmi_match_plan.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    L.siftmi_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.siftmi_host_free.argtypes = [vp]
    L.siftmi_approximate_match.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_float, C.c_float, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.siftmi_match_geometry.argtypes = [vp, vp, vp, C.c_int64, vp, vp, C.c_int64, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
    L.siftmi_descriptor_index.argtypes = [vp, C.c_int64, vp, vp, vp]
    L.siftmi_descriptor_to_reference.argtypes = [vp, C.c_int64, vp]
    L.siftmi_descriptor_to_reference.restype = None
    L.siftmi_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.siftmi_octave_size.argtypes = [vp, C.c_int, i32p, i32p, C.POINTER(C.c_float)]
    L.siftmi_get_sigma.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.siftmi_get_weights.argtypes = [vp, C.c_int, vp, i32p]
    L.siftmi_copy_gaussian.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    L.siftmi_copy_dog.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    L.siftmi_copy_extrema.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int32, i32p]
    L.siftmi_copy_orientations.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int32, i32p]
    L.siftmi_copy_descriptor_floats.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int32, i32p]
    L.siftmi_enable_timings.argtypes = [vp, C.c_int]
    L.siftmi_reset_timings.argtypes = [vp]
    L.siftmi_get_timings.argtypes = [vp, vp, vp]
    L.siftmi_get_blur_layer_timings.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), i32p]
    L.siftmi_blur_algorithmic_bytes.argtypes = [vp, C.c_int]
    L.siftmi_blur_algorithmic_bytes.restype = C.c_int64
    L.siftmi_time_blur.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.siftmi_time_copy.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.siftmi_time_blur_memory.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.siftmi_synchronize.argtypes = [vp]
    i64p = C.POINTER(C.c_int64)
    L.siftmi_device_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(vp)]
    L.siftmi_device_free.argtypes = [vp]
    L.siftmi_memcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
    L.siftmi_device_synchronize.argtypes = [C.c_int]
    L.siftmi_stream_default_config.argtypes = [C.POINTER(StreamConfig), C.c_int32]
    L.siftmi_stream_create.argtypes = [vp, C.POINTER(StreamConfig), C.POINTER(vp)]
    L.siftmi_stream_destroy.argtypes = [vp]
    L.siftmi_stream_destroy.restype = None
    L.siftmi_stream_context.argtypes = [vp, C.c_int]
    L.siftmi_stream_context.restype = vp
    L.siftmi_stream_submit_device.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, i64p]
    L.siftmi_stream_submit_host.argtypes = [vp, vp, C.c_size_t, C.c_size_t, i64p]
    L.siftmi_stream_wait_upload.argtypes = [vp, C.c_int64]
    L.siftmi_stream_wait_consumed.argtypes = [vp, C.c_int64]
    L.siftmi_stream_result_device.argtypes = [vp, C.c_int, C.POINTER(StepDevice), vp]
    L.siftmi_stream_result_host.argtypes = [vp, C.c_int, C.POINTER(StepHost)]
    L.siftmi_stream_synchronize.argtypes = [vp]
    L.siftmi_stream_set_density_mode.argtypes = [vp, C.c_int]
    L.siftmi_graph_stats.argtypes = [vp, i64p, i64p, i64p, i32p]
    L.siftmi_exchange_ranks.argtypes = [vp, i32p, i32p]
    L.siftmi_exchange_set_timeout.argtypes = [vp, C.c_double]
    L.siftmi_exchange_wait.argtypes = [vp]
    L.siftmi_exchange_unique_id.argtypes = [vp]
    L.siftmi_exchange_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.siftmi_exchange_destroy.argtypes = [vp]
    L.siftmi_exchange_destroy.restype = None
    L.siftmi_exchange_gather.argtypes = [vp, C.c_int]
    L.siftmi_exchange_result.argtypes = [vp, C.c_int, C.POINTER(Gathered), vp, C.c_int]
    L.siftmi_exchange_finish.argtypes = [vp, i64p, i64p]
    L.siftmi_exchange_set_headroom.argtypes = [vp, C.c_int32, C.c_int64]
    L.siftmi_exchange_stats.argtypes = [vp, C.POINTER(C.c_double), i64p, i64p]
    L.siftmi_exchange_transport.argtypes = []
    L.siftmi_exchange_transport.restype = C.c_char_p
    L.siftmi_gather_plan_init.argtypes = [C.POINTER(GatherPlan), C.c_int64, C.c_int64]
    L.siftmi_gather_plan_resolve.argtypes = [C.POINTER(GatherPlan), vp, C.c_int, C.c_int64, C.c_int64]
    _lib = L
    return L


def check(rc, allow_capacity=False):
    if rc == OK or (allow_capacity and rc == E_CAPACITY):
        return rc
    raise SiftmiError(rc, load().siftmi_last_error().decode("utf-8", "replace"))


def default_config(width, height, **overrides):
    cfg = Config()
    check(load().siftmi_default_config(C.byref(cfg), width, height))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError("siftmi_config has no field %r" % k)
        setattr(cfg, k, v)
    return cfg
