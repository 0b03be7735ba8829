"""Batched frame stream on one GPU: frames and results stay in HBM (torch tensors own the memory), optional RCCL
all-gather of the packed results.  The HIP path runs on a launch stream owned by the FrameStream, ordered after torch's
current stream on entry and before it on exit, so torch work issued before run() (the frame upload) and after it
(`.cpu()`, collectives) is ordered with the kernels whatever torch's current stream is -- including the default stream,
whose handle is NULL and would otherwise select the context's own, unordered stream.

With `overlap_gather` (the multi-GPU driver) the packed results alternate between TWO buffer sets and the all-gather of
step k runs on a side stream: the kernels of step k+1 write the other set while RCCL reads this one over xGMI, so the
exchange (7 peers x ~30 MB per rank and step on the fully connected mesh: a few ms, per-link bound) hides under compute
instead of adding to every step.  A set is reused two steps later, after its gather has finished (event).

With `pipeline=2` consecutive steps alternate between TWO contexts (the engine and a clone: two pyramids, two launch
streams, two result sets).  A step is HBM-bound for its first three quarters (pyramid, extrema) and VALU-bound for the rest
(orientation, descriptors); with two steps in flight the second's dense stages run under the first's keypoint stages
(measured on MI355X, 64 x 1080p: 11.5 -> 10.5 ms per step; tools/overlap_experiment.py).  run() then no longer orders
torch's current stream after the step: results_host(), all_gather() and wait() do, for the step they read."""
import numpy as np
import torch

from . import _capi, dist as smdist


class _ResultSet:
    def __init__(self, kp_cap, desc_cap, frames, n_octaves, device):
        self.kp = torch.empty(kp_cap * smdist.KP_BYTES, dtype=torch.uint8, device=device)
        self.desc = torch.empty(desc_cap * smdist.DESC_BYTES, dtype=torch.uint8, device=device)
        self.counts = torch.zeros((2, frames, n_octaves), dtype=torch.int32, device=device)
        self.totals = torch.zeros(4, dtype=torch.int32, device=device)      # {n_kp, n_desc, overflow flags, 0}
        self.gather_done = None                                             # event: the side-stream gather that read this set
        self.ready = None                                                   # event: the step that wrote this set (launch stream)


class FrameStream:
    def __init__(self, engine, frames_per_step, device, world_size=1, kp_per_frame=32768, desc_per_frame=49152, overlap_gather=False,
                 pipeline=1, result_sets=None):
        assert 1 <= pipeline <= 4
        self.eng = engine
        self.engines = [engine] + [engine.clone() for _ in range(pipeline - 1)]
        self.pipeline = pipeline
        self.F = frames_per_step
        self.device = device
        self.world = world_size
        self.kp_cap = kp_per_frame * frames_per_step
        self.desc_cap = desc_per_frame * frames_per_step
        self.overlap = bool(overlap_gather)
        # a multiple of the number of contexts, so that a context always meets the same result sets: the library replays a
        # captured launch sequence per (input, output, stream) signature, and every new pairing would be captured afresh
        n_sets = max(pipeline, 2 if self.overlap else 1, result_sets or 1)
        n_sets = (n_sets + pipeline - 1) // pipeline * pipeline
        self.sets = [_ResultSet(self.kp_cap, self.desc_cap, frames_per_step, engine.n_octaves, device) for _ in range(n_sets)]
        self.cur = 0                                         # the set the last run() wrote
        self.step_no = -1
        self.gathered = None
        self.exchange = smdist.ResultExchange(self.kp_cap, self.desc_cap)
        self.launch_streams = [torch.cuda.Stream(device=device) for _ in self.engines]
        self.launch_stream = self.launch_streams[0]          # the stream of the last run()
        self.gather_stream = torch.cuda.Stream(device=device) if self.overlap else None
        self.gather_events = []                              # (start, end) timing events of every all_gather() call
        self.copy_stream = None                              # run_host(): uploads, staging buffers and the step that last read each
        self._staging, self._staging_read = [], []
        self._uploaded = None

    def close(self):
        """Destroy the contexts this stream created (the clones of pipeline > 1); the engine it was given stays the caller's."""
        torch.cuda.synchronize(self.device)
        for e in self.engines[1:]:
            e.close()
        self.engines = self.engines[:1]
        self.pipeline = 1

    # the buffers of the last step (what results_host / all_gather read)
    @property
    def kp(self):
        return self.sets[self.cur].kp

    @property
    def desc(self):
        return self.sets[self.cur].desc

    @property
    def counts(self):
        return self.sets[self.cur].counts

    @property
    def totals(self):
        return self.sets[self.cur].totals

    def run(self, d_frames):
        """d_frames: uint8 [F, H, W, 4] (BGRA) / [F, H, W] (gray) or float32 [F, H, W] device tensor."""
        assert d_frames.is_cuda and d_frames.shape[0] == self.F and d_frames.is_contiguous()
        if d_frames.dtype == torch.uint8 and d_frames.dim() == 4:
            fmt = _capi.FMT_BGRA8
        elif d_frames.dtype == torch.uint8:
            fmt = _capi.FMT_GRAY8
        else:
            fmt = _capi.FMT_GRAYF32
        es = d_frames.element_size()
        self.step_no += 1
        self.cur = self.step_no % len(self.sets)
        eng = self.engines[self.step_no % len(self.engines)]          # pipeline > 1: contexts and result sets rotate together
        self.launch_stream = self.launch_streams[self.step_no % len(self.engines)]
        rs = self.sets[self.cur]
        cur = torch.cuda.current_stream(self.device)
        self.launch_stream.wait_stream(cur)
        if rs.gather_done is not None:                       # the gather that read this set two steps ago
            self.launch_stream.wait_event(rs.gather_done)
        if self._uploaded is not None:                       # run_host(): the frames arrive on the copy stream
            self.launch_stream.wait_event(self._uploaded)
            self._uploaded = None
        d_frames.record_stream(self.launch_stream)
        eng.detect_describe_batch_device(self.F, d_frames.data_ptr(), fmt, d_frames.stride(1) * es, d_frames.stride(0) * es,
                                         rs.kp.data_ptr(), self.kp_cap, rs.desc.data_ptr(), self.desc_cap,
                                         rs.counts.data_ptr(), rs.totals.data_ptr(), self.launch_stream.cuda_stream)
        rs.ready = torch.cuda.Event()
        rs.ready.record(self.launch_stream)
        if self.pipeline == 1:
            cur.wait_stream(self.launch_stream)
        # pipeline > 1: the next step must be able to start before this one ends, so nothing joins here; readers call wait()

    def run_host(self, h_frames):
        """One step on frames in PAGE-LOCKED host memory (a pinned torch tensor, layouts as run()).  The upload goes to one of
        a few (more than `pipeline`) staging buffers on a copy stream, ordered after the step that last read that buffer only: the PCIe
        transfer of step k+1 runs under the kernels of step k.  Read results with results_host(previous=True) after
        launching the next step to keep both engines busy."""
        assert (not h_frames.is_cuda) and h_frames.is_pinned() and h_frames.shape[0] == self.F and h_frames.is_contiguous()
        # (Letting the seed kernel read the page-locked frames over PCIe itself instead of staging them: 23.8 against 13.5-15.3 ms
        # per 64 x 1080p step.)
        if self.copy_stream is None:
            self.copy_stream = torch.cuda.Stream(device=self.device)
            # more buffers than steps in flight, so that an upload never waits for a running step; as many as result sets when
            # that is enough: buffer, result set and context then rotate together (one launch signature per buffer)
            n = len(self.sets) if len(self.sets) > self.pipeline else 2 * self.pipeline
            self._staging = [torch.empty(h_frames.shape, dtype=h_frames.dtype, device=self.device) for _ in range(n)]
            self._staging_read = [None] * n
        slot = (self.step_no + 1) % len(self._staging)
        st = self._staging[slot]
        assert st.shape == h_frames.shape and st.dtype == h_frames.dtype, "run_host: one frame layout per stream"
        if self._staging_read[slot] is not None:
            self.copy_stream.wait_event(self._staging_read[slot])
        with torch.cuda.stream(self.copy_stream):
            st.copy_(h_frames, non_blocking=True)
            self._uploaded = torch.cuda.Event()
            self._uploaded.record(self.copy_stream)
        self.run(st)
        self._staging_read[slot] = self.sets[self.cur].ready

    def wait(self, previous=False, back=None):
        """Order torch's current stream after the last step -- or the one `back` steps before it -- (no host synchronisation)."""
        back = (1 if previous else 0) if back is None else back
        rs = self.sets[(self.step_no - back) % len(self.sets)]
        if rs.ready is not None:
            torch.cuda.current_stream(self.device).wait_event(rs.ready)

    def all_gather(self, synchronous=False):
        """RCCL all-gather of the last step's packed results.  Default: payload sizes come from the previous step's counts, so
        nothing synchronises the host inside the step (ResultExchange); synchronous=True sizes them from this step's.
        With overlap_gather the collectives run on the side stream: call wait_gather() (or synchronise the device) before
        reading the returned tensors."""
        rs = self.sets[self.cur]
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if self.overlap and not synchronous:
            self.gather_stream.wait_event(rs.ready)
            with torch.cuda.stream(self.gather_stream):
                t0.record()
                self.gathered = self.exchange.gather(rs.kp, rs.desc, rs.counts, rs.totals)
                t1.record()
                rs.gather_done = torch.cuda.Event()
                rs.gather_done.record(self.gather_stream)
        else:
            self.wait()
            t0.record()
            if synchronous:
                self.gathered = smdist.gather_results(rs.kp, rs.desc, rs.counts, rs.totals)
            else:
                self.gathered = self.exchange.gather(rs.kp, rs.desc, rs.counts, rs.totals)
            t1.record()
        self.gather_events.append((t0, t1))
        del self.gather_events[:-64]
        return self.gathered

    def wait_gather(self):
        """Order torch's current stream after the side-stream gather (no host synchronisation)."""
        if self.overlap:
            torch.cuda.current_stream(self.device).wait_stream(self.gather_stream)

    def results_host(self, allow_capacity=False, previous=False, back=None):
        """Packed results of the last step on the host; back=n (previous=True: n = 1): of the step n before it, which a
        pipelined consumer reads while the later ones are still running (needs more than n result sets:
        FrameStream(result_sets=...))."""
        back = (1 if previous else 0) if back is None else back
        assert 0 <= back < len(self.sets) and self.step_no >= back
        rs = self.sets[(self.step_no - back) % len(self.sets)]
        self.wait(back=back)
        tot = rs.totals.cpu().numpy()
        nk, nd = int(tot[0]), int(tot[1])
        if tot[2] and not allow_capacity:       # the condition the host-facing API reports as SIFTMI_E_CAPACITY
            raise _capi.SiftmiError(_capi.E_CAPACITY, "list capacity exceeded on the device path (overflow flags 0x%x): results truncated" % int(tot[2]))
        kp = rs.kp[:nk * smdist.KP_BYTES].cpu().numpy().view(_capi.keypoint_dtype)
        ds = rs.desc[:nd * smdist.DESC_BYTES].cpu().numpy().view(_capi.descriptor_dtype)
        return {"n_keypoints": nk, "n_descriptors": nd, "keypoints": kp, "descriptors": ds, "counts": rs.counts.cpu().numpy(),
                "overflow_flags": int(tot[2])}
