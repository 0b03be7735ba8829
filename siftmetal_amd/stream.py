"""Batched frame stream on one GPU: frames and results stay in HBM (torch tensors own the memory), optional RCCL
all-gather of the packed results.  The HIP path runs on a launch stream owned by the FrameStream, ordered after torch's
current stream on entry and before it on exit, so torch work issued before run() (the frame upload) and after it
(`.cpu()`, collectives) is ordered with the kernels whatever torch's current stream is -- including the default stream,
whose handle is NULL and would otherwise select the context's own, unordered stream."""
import numpy as np
import torch

from . import _capi, dist as smdist


class FrameStream:
    def __init__(self, engine, frames_per_step, device, world_size=1, kp_per_frame=32768, desc_per_frame=49152):
        self.eng = engine
        self.F = frames_per_step
        self.device = device
        self.world = world_size
        self.kp_cap = kp_per_frame * frames_per_step
        self.desc_cap = desc_per_frame * frames_per_step
        self.kp = torch.empty(self.kp_cap * smdist.KP_BYTES, dtype=torch.uint8, device=device)
        self.desc = torch.empty(self.desc_cap * smdist.DESC_BYTES, dtype=torch.uint8, device=device)
        self.counts = torch.zeros((2, frames_per_step, engine.n_octaves), dtype=torch.int32, device=device)
        self.totals = torch.zeros(4, dtype=torch.int32, device=device)      # {n_kp, n_desc, overflow flags, 0}
        self.gathered = None
        self.exchange = smdist.ResultExchange(self.kp_cap, self.desc_cap)
        self.launch_stream = torch.cuda.Stream(device=device)

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
        cur = torch.cuda.current_stream(self.device)
        self.launch_stream.wait_stream(cur)
        d_frames.record_stream(self.launch_stream)
        self.eng.detect_describe_batch_device(self.F, d_frames.data_ptr(), fmt, d_frames.stride(1) * es, d_frames.stride(0) * es,
                                              self.kp.data_ptr(), self.kp_cap, self.desc.data_ptr(), self.desc_cap,
                                              self.counts.data_ptr(), self.totals.data_ptr(), self.launch_stream.cuda_stream)
        cur.wait_stream(self.launch_stream)

    def all_gather(self, synchronous=False):
        """RCCL all-gather of this step's packed results.  Default: payload sizes come from the previous step's counts, so
        nothing synchronises the host inside the step (ResultExchange); synchronous=True sizes them from this step's."""
        if synchronous:
            self.gathered = smdist.gather_results(self.kp, self.desc, self.counts, self.totals)
        else:
            self.gathered = self.exchange.gather(self.kp, self.desc, self.counts, self.totals)
        return self.gathered

    def results_host(self, allow_capacity=False):
        tot = self.totals.cpu().numpy()
        nk, nd = int(tot[0]), int(tot[1])
        if tot[2] and not allow_capacity:       # the condition the host-facing API reports as SIFTMI_E_CAPACITY
            raise _capi.SiftmiError(_capi.E_CAPACITY, "list capacity exceeded on the device path (overflow flags 0x%x): results truncated" % int(tot[2]))
        kp = self.kp[:nk * smdist.KP_BYTES].cpu().numpy().view(_capi.keypoint_dtype)
        ds = self.desc[:nd * smdist.DESC_BYTES].cpu().numpy().view(_capi.descriptor_dtype)
        return {"n_keypoints": nk, "n_descriptors": nd, "keypoints": kp, "descriptors": ds, "counts": self.counts.cpu().numpy(),
                "overflow_flags": int(tot[2])}
