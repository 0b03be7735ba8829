"""ctypes binding of the C ABI's frame stream and result exchange (include/siftmi.h: siftmi_stream_*, siftmi_exchange_*).

Everything that makes the stream fast lives in libsiftmi.so: consecutive steps alternating between contexts on their own
launch streams (`pipeline` steps in flight), rotating result sets, host frames uploaded on a copy stream into rotating
staging buffers, results copied back on a third stream, and the RCCL all-gather of every rank's packed results on a side
stream (librccl, loaded by the library itself).  This module only passes pointers: frames may be torch CUDA tensors (their
`data_ptr()`; the kernels are ordered after torch's current stream), `DeviceFrames` (HBM allocated through the C ABI, no
torch) or page-locked host arrays (run_host).  A Swift host binds the same entry points (swift/)."""
import ctypes as C

import numpy as np

from . import _capi

KP_BYTES, DESC_BYTES = 44, 136


def shard_frames(n_frames, world_size, rank):
    """Frame-per-GPU sharding of a frame stream: frame i -> rank i mod world_size; this rank's frame indices.
    Frames are independent units (the reference keeps no cross-frame state), so nothing else is partitioned."""
    return list(range(rank, n_frames, world_size))


def _torch_stream_handle(device):
    """hipStream_t of torch's current stream on `device` (0 = the legacy default stream) -- torch is only touched when the
    caller hands in torch tensors."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


class DeviceFrames:
    """Frames resident in HBM, allocated and filled through the C ABI (siftmi_device_alloc / siftmi_memcpy)."""

    def __init__(self, array, device=0):
        a = np.ascontiguousarray(array)
        self.shape, self.dtype, self.strides, self.nbytes = a.shape, a.dtype, a.strides, a.nbytes
        self.device = device
        p = C.c_void_p()
        _capi.check(_capi.load().siftmi_device_alloc(device, a.nbytes, C.byref(p)))
        self.ptr = p.value
        _capi.check(_capi.load().siftmi_memcpy(self.ptr, a.ctypes.data, a.nbytes, 0))

    def data_ptr(self):
        return self.ptr

    def close(self):
        if getattr(self, "ptr", None):
            _capi.load().siftmi_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _format_of(dtype_is_u8, ndim):
    if dtype_is_u8 and ndim == 4:
        return _capi.FMT_BGRA8
    if dtype_is_u8:
        return _capi.FMT_GRAY8
    return _capi.FMT_GRAYF32


def _describe(frames):
    """(pointer, format, row stride, frame stride, frames, is_torch) of a frame batch."""
    if hasattr(frames, "is_cuda"):                           # torch tensor
        import torch
        assert frames.is_contiguous()
        es = frames.element_size()
        fmt = _format_of(frames.dtype == torch.uint8, frames.dim())
        return frames.data_ptr(), fmt, frames.stride(1) * es, frames.stride(0) * es, frames.shape[0], True
    if isinstance(frames, DeviceFrames):
        fmt = _format_of(frames.dtype == np.uint8, len(frames.shape))
        return frames.ptr, fmt, frames.strides[1], frames.strides[0], frames.shape[0], False
    a = frames                                               # numpy array over page-locked memory
    fmt = _format_of(a.dtype == np.uint8, a.ndim)
    return a.ctypes.data, fmt, a.strides[1], a.strides[0], a.shape[0], False


class _BorrowedEngine:
    """Introspection handle on a context the stream owns (its clones); close() is the stream's business."""

    def __new__(cls, engine, handle):
        from . import Engine
        e = Engine.__new__(Engine)
        e.L, e.cfg, e.h = engine.L, engine.cfg, C.c_void_p(handle)
        e.width, e.height, e.device = engine.width, engine.height, engine.device
        e.n_octaves, e.nspo, e.max_batch = engine.n_octaves, engine.nspo, engine.max_batch
        e.close = lambda: None
        return e


class Exchange:
    """siftmi_exchange_*: RCCL all-gather of the stream's packed results (one communicator rank per process)."""

    def __init__(self, stream, rank=0, world=1, unique_id=None):
        self.L = _capi.load()
        self.stream, self.rank, self.world = stream, rank, world
        if unique_id is None:
            unique_id = self.make_unique_id() if world == 1 else None
        assert unique_id is not None and len(unique_id) == _capi.UNIQUE_ID_BYTES, "every rank needs rank 0's unique id"
        h = C.c_void_p()
        buf = (C.c_char * _capi.UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _capi.check(self.L.siftmi_exchange_create(stream.h, buf, rank, world, C.byref(h)))
        self.h = h

    @staticmethod
    def make_unique_id():
        buf = (C.c_char * _capi.UNIQUE_ID_BYTES)()
        _capi.check(_capi.load().siftmi_exchange_unique_id(buf))
        return bytes(buf)

    def ranks(self):
        """(ranks, this rank) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int32(), C.c_int32()
        _capi.check(self.L.siftmi_exchange_ranks(self.h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def set_timeout(self, seconds):
        """Deadline of every host wait of the exchange; on expiry the communicator is aborted and the call raises."""
        _capi.check(self.L.siftmi_exchange_set_timeout(self.h, float(seconds)))

    def wait(self):
        """Bounded host wait for every collective enqueued so far (call before an unbounded device synchronisation)."""
        _capi.check(self.L.siftmi_exchange_wait(self.h))

    def set_headroom(self, percent, quantum=1024):
        _capi.check(self.L.siftmi_exchange_set_headroom(self.h, int(percent), int(quantum)))

    def gather(self, synchronous=False):
        _capi.check(self.L.siftmi_exchange_gather(self.h, int(bool(synchronous))))

    def result(self, back=0, consumer_stream=_capi.NO_STREAM, wait_host=False):
        g = _capi.Gathered()
        _capi.check(self.L.siftmi_exchange_result(self.h, back, C.byref(g), consumer_stream, int(bool(wait_host))))
        return g

    def result_host(self, back=0):
        """The gathered step on the host: per rank the valid keypoint / descriptor records and the counts (blocks)."""
        g = self.result(back, wait_host=True)
        F, O = self.stream.F, self.stream.n_octaves
        tot = np.ctypeslib.as_array(g.totals_host, shape=(g.world, 4)).copy()
        counts = np.empty((g.world, 2, F, O), np.int32)
        _capi.check(self.L.siftmi_memcpy(counts.ctypes.data, g.counts, counts.nbytes, 1))
        kps, descs = [], []
        for r in range(g.world):
            nk, nd = min(int(tot[r, 0]), g.kp_records), min(int(tot[r, 1]), g.desc_records)
            k, d = np.empty(nk, _capi.keypoint_dtype), np.empty(nd, _capi.descriptor_dtype)
            if nk:
                _capi.check(self.L.siftmi_memcpy(k.ctypes.data, g.keypoints + r * g.kp_stride, k.nbytes, 1))
            if nd:
                _capi.check(self.L.siftmi_memcpy(d.ctypes.data, g.descriptors + r * g.desc_stride, d.nbytes, 1))
            kps.append(k)
            descs.append(d)
        return {"step": int(g.step), "complete": bool(g.complete), "totals": tot, "counts": counts, "keypoints": kps, "descriptors": descs,
                "records_per_rank": (int(g.kp_records), int(g.desc_records))}

    def finish(self):
        """Collective, end of stream: (steps that had to be gathered again in full, steps with list overflow on some rank)."""
        a, b = C.c_int64(), C.c_int64()
        _capi.check(self.L.siftmi_exchange_finish(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def stats(self):
        ms, n, nb = C.c_double(), C.c_int64(), C.c_int64()
        _capi.check(self.L.siftmi_exchange_stats(self.h, C.byref(ms), C.byref(n), C.byref(nb)))
        return {"ms": ms.value, "gathers": int(n.value), "bytes_last": int(nb.value)}

    def close(self):
        if getattr(self, "h", None):
            self.L.siftmi_exchange_destroy(self.h)
            self.h = None


class FrameStream:
    """siftmi_stream_* on one GPU.  pipeline = steps in flight (contexts); result_sets rotating output buffer sets."""

    def __init__(self, engine, frames_per_step, device=None, world_size=1, kp_per_frame=32768, desc_per_frame=49152, overlap_gather=False,
                 pipeline=1, result_sets=None, fmt=_capi.FMT_BGRA8, rank=0, unique_id=None, density_mode=0):
        assert 1 <= pipeline <= 4
        self._density_mode = density_mode
        self.L = _capi.load()
        self.eng, self.F, self.pipeline = engine, frames_per_step, pipeline
        self.n_octaves = engine.n_octaves
        self.device = device if device is not None else engine.device
        self.world, self.rank = world_size, rank
        self._kp_per_frame, self._desc_per_frame = kp_per_frame, desc_per_frame
        self._result_sets = result_sets if result_sets else max(pipeline, 2 if overlap_gather else 1)
        self.h = None
        self._create(fmt)
        self.exchange = Exchange(self, rank, world_size, unique_id) if overlap_gather else None
        self.step_no = -1

    def _create(self, fmt):
        scfg = _capi.StreamConfig()
        _capi.check(self.L.siftmi_stream_default_config(C.byref(scfg), self.F))
        scfg.steps_in_flight, scfg.result_sets, scfg.format = self.pipeline, self._result_sets, fmt
        scfg.kp_per_frame, scfg.desc_per_frame = self._kp_per_frame, self._desc_per_frame
        scfg.density_mode = self._density_mode
        h = C.c_void_p()
        _capi.check(self.L.siftmi_stream_create(self.eng.h, C.byref(scfg), C.byref(h)))
        self.h, self.fmt = h, fmt
        self.engines = [self.eng] + [_BorrowedEngine(self.eng, self.L.siftmi_stream_context(h, i)) for i in range(1, self.pipeline)]
        n_sets = max(self._result_sets, self.pipeline)
        self.n_sets = (n_sets + self.pipeline - 1) // self.pipeline * self.pipeline
        self.sets = [None] * self.n_sets                     # the input tensors of the steps still in flight (kept alive)
        self.kp_cap, self.desc_cap = self._kp_per_frame * self.F, self._desc_per_frame * self.F

    def close(self):
        """Destroys the stream and the contexts it created (the clones of pipeline > 1); the engine it was given stays the caller's."""
        if getattr(self, "exchange", None):
            self.exchange.close()
            self.exchange = None
        if getattr(self, "h", None):
            self.L.siftmi_stream_destroy(self.h)
            self.h = None
        self.engines = self.engines[:1]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _hold(self, frames):
        # A torch tensor handed to run() may be a temporary: torch's allocator knows nothing of the library's streams and would
        # recycle its memory while the kernels still read it.  Keep it until its result set is reused, and then only let go
        # once the step that read it has run.
        idx = (self.step_no + 1) % self.n_sets
        if self.sets[idx] is not None:
            _capi.check(self.L.siftmi_stream_wait_consumed(self.h, self.step_no + 1 - self.n_sets))
        self.sets[idx] = frames

    def run(self, d_frames):
        """One step on frames resident in HBM: uint8 [F, H, W, 4] (BGRA) / [F, H, W] (gray) or float32 [F, H, W]; a torch CUDA
        tensor (ordered after torch's current stream) or DeviceFrames.  Asynchronous."""
        ptr, fmt, row, frame, n, is_torch = _describe(d_frames)
        assert n == self.F
        if fmt != self.fmt:
            assert self.step_no < 0 and self.exchange is None, "one frame format per stream"
            self.L.siftmi_stream_destroy(self.h)
            self._create(fmt)
        producer = _torch_stream_handle(self.device) if is_torch else _capi.NO_STREAM
        self._hold(d_frames)
        step = C.c_int64()
        _capi.check(self.L.siftmi_stream_submit_device(self.h, ptr, row, frame, producer, C.byref(step)))
        self.step_no = step.value
        if is_torch and self.pipeline == 1:                  # torch work issued after run() sees the results (one step in flight only:
            self.wait()                                      # with more, joining here would serialise the steps; readers call wait())
        return self.step_no

    def run_host(self, h_frames):
        """One step on frames in PAGE-LOCKED host memory (a pinned torch tensor or a pinned_empty numpy array, layouts as
        run()).  The upload of step k+1 runs under the kernels of step k.  `h_frames` must stay untouched until
        wait_upload(step) returns (or the step's results have been read)."""
        if hasattr(h_frames, "is_pinned"):
            assert (not h_frames.is_cuda) and h_frames.is_pinned() and h_frames.is_contiguous()
            a = h_frames.numpy()
        else:
            a = h_frames
        ptr, fmt, row, frame, n, _ = _describe(a)
        assert n == self.F
        if fmt != self.fmt:
            assert self.step_no < 0 and self.exchange is None, "one frame format per stream"
            self.L.siftmi_stream_destroy(self.h)
            self._create(fmt)
        self._hold(h_frames)
        step = C.c_int64()
        _capi.check(self.L.siftmi_stream_submit_host(self.h, ptr, row, frame, C.byref(step)))
        self.step_no = step.value
        return self.step_no

    def wait_upload(self, step=None):
        """Blocks until the frames of `step` (default: the last run_host) have left host memory: the caller may refill them."""
        _capi.check(self.L.siftmi_stream_wait_upload(self.h, self.step_no if step is None else step))

    def result_device(self, back=0, consumer_stream=_capi.NO_STREAM):
        r = _capi.StepDevice()
        _capi.check(self.L.siftmi_stream_result_device(self.h, back, C.byref(r), consumer_stream))
        return r

    def wait(self, previous=False, back=None):
        """Order torch's current stream after the last step -- or the one `back` steps before it -- (no host synchronisation)."""
        back = (1 if previous else 0) if back is None else back
        self.result_device(back, _torch_stream_handle(self.device))

    def synchronize(self):
        _capi.check(self.L.siftmi_stream_synchronize(self.h))

    def set_density_mode(self, mode):
        """0: the launch sequence of a step is chosen from earlier steps' descriptor totals; 1: always the sparse form (forked,
        activity flags); 2: always the dense form (one chain, full extrema scan).  The records do not depend on it."""
        _capi.check(self.L.siftmi_stream_set_density_mode(self.h, int(mode)))
        self._density_mode = int(mode)

    def all_gather(self, synchronous=False):
        """RCCL all-gather of the last step's packed results on the exchange's side stream (siftmi_exchange_gather): payload
        sizes from the previous step's counts, no host synchronisation; a step that turns out to have been cut short is
        gathered again in full by the next call (or finish()).  Read with exchange.result()/result_host()."""
        self.exchange.gather(synchronous)

    def results_host(self, allow_capacity=False, previous=False, back=None, copy=True):
        """Packed results of the last step on the host; back=n (previous=True: n = 1): of the step n before it, which a
        pipelined consumer reads while the later ones are still running (needs more than n result sets).  copy=False returns
        views of the stream's page-locked buffers, valid until that result set is reused (result_sets steps later)."""
        back = (1 if previous else 0) if back is None else back
        r = _capi.StepHost()
        _capi.check(self.L.siftmi_stream_result_host(self.h, back, C.byref(r)), allow_capacity=True)
        if r.overflow_flags & 32:                            # a float frame outside [0, 1]: results unusable (siftmi_format)
            raise _capi.SiftmiError(_capi.E_BADARG, "SIFTMI_FMT_GRAYF32 frame with a value outside [0, 1]")
        if r.overflow_flags and not allow_capacity:          # the condition the host-facing API reports as SIFTMI_E_CAPACITY
            raise _capi.SiftmiError(_capi.E_CAPACITY, "list capacity exceeded on the device path (overflow flags 0x%x): results truncated" % r.overflow_flags)
        nk, nd = r.n_keypoints, r.n_descriptors
        kp = np.ctypeslib.as_array(C.cast(r.keypoints, C.POINTER(C.c_uint8)), shape=(nk * KP_BYTES,)).view(_capi.keypoint_dtype) \
            if nk else np.zeros(0, _capi.keypoint_dtype)
        ds = np.ctypeslib.as_array(C.cast(r.descriptors, C.POINTER(C.c_uint8)), shape=(nd * DESC_BYTES,)).view(_capi.descriptor_dtype) \
            if nd else np.zeros(0, _capi.descriptor_dtype)
        counts = np.ctypeslib.as_array(r.counts, shape=(2, self.F, self.n_octaves))
        if copy:
            kp, ds, counts = kp.copy(), ds.copy(), counts.copy()
        return {"step": int(r.step), "n_keypoints": nk, "n_descriptors": nd, "keypoints": kp, "descriptors": ds, "counts": counts,
                "overflow_flags": int(r.overflow_flags), "launch_flags": int(r.launch_flags)}
