"""TEST-ONLY mirror of the result exchange's protocol over torch.distributed (gloo), for the CPU tests (tests/test_dist_gloo.py).

The product exchange is siftmi_exchange_* inside libsiftmi.so (siftmetal_amd/csrc/exchange_api.hip.h; multi-rank runs of it:
tests/test_exchange_ranks.py).  What this module adds is a way to drive the library's own SIZING RULE (siftmi_gather_plan_*, host
arithmetic exported by the C ABI) with world size 2 where there is no GPU.  Nothing in siftmetal_amd/ imports it.

Frame-per-GPU sharding and the descriptor all-gather (the only exchange step of the path).

Frames are independent units (the reference keeps no cross-frame state: SIFT.swift holds only
scratch), so a stream of frames shards with no data-path collective; after a batch every rank
publishes its packed keypoint / descriptor buffers with ONE count exchange + padded all-gathers
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
The functions here are device-agnostic tensor plumbing; no SIFT arithmetic.

The product exchange is siftmi_exchange_* inside libsiftmi.so (RCCL, siftmetal_amd/csrc/exchange_api.hip.h; bound by
stream.Exchange).  This module holds what runs without a GPU: the sharding rule, a one-shot gather over
torch.distributed, and ResultExchange -- the same per-step protocol as siftmi_exchange_gather over torch.distributed
(gloo in the CPU tests), sized by the library's own rule (siftmi_gather_plan_*), so that the rule and the protocol
(step-late sizing, re-gather of a step that was cut short) are covered on CPU with world size 2.

  gather_results   one-shot, sizes the payload gathers from the count exchange (one host sync).
  ResultExchange   for a frame stream: the payload gathers of step k are sized from the counts of
                   step k-1 (plus headroom), and the counts of step k are read on the host while step
                   k+1 runs -- no host synchronisation between a step's kernels and its collectives.
                   A step whose counts turn out to exceed what was sent is gathered again in full by the
                   next gather() / finish() call, while its result buffers are still intact.
"""
import ctypes as C
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

KP_BYTES, DESC_BYTES = 44, 136
TOTALS = 4                      # {n_keypoints, n_descriptors, overflow_flags, 0} (include/siftmi.h, d_totals)


from siftmetal_amd.stream import shard_frames  # noqa: E402,F401  (the sharding rule itself is product code)


def _totals_row(totals: torch.Tensor) -> torch.Tensor:
    """[1, TOTALS] int32 from a 2- or 4-element totals tensor."""
    t = totals.reshape(-1)
    if t.numel() < TOTALS:
        t = torch.cat([t, torch.zeros(TOTALS - t.numel(), dtype=t.dtype, device=t.device)])
    return t[:TOTALS].reshape(1, TOTALS).contiguous()


def _payload_gathers(kp_bytes, desc_bytes, counts, n_kp, n_desc, group):
    world = dist.get_world_size(group)
    all_counts = torch.empty((world,) + tuple(counts.shape), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(all_counts, counts.contiguous().unsqueeze(0), group=group)
    all_kp = torch.empty((world, n_kp * KP_BYTES), dtype=torch.uint8, device=kp_bytes.device)
    dist.all_gather_into_tensor(all_kp, kp_bytes[:n_kp * KP_BYTES].unsqueeze(0), group=group)
    all_desc = torch.empty((world, n_desc * DESC_BYTES), dtype=torch.uint8, device=desc_bytes.device)
    dist.all_gather_into_tensor(all_desc, desc_bytes[:n_desc * DESC_BYTES].unsqueeze(0), group=group)
    return all_counts, all_kp, all_desc


def gather_results(kp_bytes: torch.Tensor, desc_bytes: torch.Tensor, counts: torch.Tensor, totals: torch.Tensor,
                   group=None) -> Dict[str, object]:
    """All-gather one batch's packed results.

    kp_bytes   uint8 [kp_capacity * 44]    packed siftmi_keypoint records (first totals[0] valid)
    desc_bytes uint8 [desc_capacity * 136] packed siftmi_descriptor records (first totals[1] valid)
    counts     int32 [2, n_frames, n_octaves]
    totals     int32 [2] or [4] = {n_keypoints, n_descriptors(, overflow_flags, 0)}
    Returns per-rank views: {"totals": [world,4] (host), "counts": [world,2,F,O], "keypoints": [world, max_kp*44],
    "descriptors": [world, max_desc*136]} -- rank r's valid bytes are the first totals[r]*record_size of row r.
    """
    world = dist.get_world_size(group)
    all_totals = torch.empty((world, TOTALS), dtype=torch.int32, device=totals.device)
    dist.all_gather_into_tensor(all_totals, _totals_row(totals), group=group)
    host_totals = all_totals.cpu()                       # the one host sync of the exchange
    max_kp = max(int(host_totals[:, 0].max()), 1)
    max_desc = max(int(host_totals[:, 1].max()), 1)
    if max_kp * KP_BYTES > kp_bytes.numel() or max_desc * DESC_BYTES > desc_bytes.numel():
        raise RuntimeError("gather_results: a peer holds more records than this rank's buffer capacity")
    all_counts, all_kp, all_desc = _payload_gathers(kp_bytes, desc_bytes, counts, max_kp, max_desc, group)
    return {"totals": host_totals, "counts": all_counts, "keypoints": all_kp, "descriptors": all_desc}


class ResultExchange:
    """Per-step all-gather of a frame stream's packed results without a host synchronisation inside the step: the protocol of
    siftmi_exchange_gather (include/siftmi.h) over torch.distributed, sized by siftmi_gather_plan_*."""

    def __init__(self, kp_capacity: int, desc_capacity: int, group=None, headroom: float = 1.25, quantum: int = 1024):
        from siftmetal_amd import _capi
        self._capi = _capi
        self.L = _capi.load()
        self.plan = _capi.GatherPlan()
        _capi.check(self.L.siftmi_gather_plan_init(C.byref(self.plan), int(kp_capacity), int(desc_capacity)))
        self.plan.headroom_percent = int(round((float(headroom) - 1.0) * 100))
        self.plan.quantum = int(quantum)
        self.group = group
        self._pending = None                             # the previous step, not yet checked against what was sent for it
        self.step = 0
        self.regathered_steps: List[int] = []            # steps whose payload gathers were too small and were gathered again in full
        self.overflow_steps: List[int] = []              # steps in which some rank reported list overflow (d_totals[2])

    def _resolve_pending(self):
        """Reads the totals of the previous step (its collectives finished long ago: the current step's kernels were enqueued
        behind them), re-gathers it in full if it was cut short, and re-sizes the payload gathers (siftmi_gather_plan_resolve)."""
        if self._pending is None:
            return
        p, self._pending = self._pending, None
        if p["event"] is not None:
            p["event"].synchronize()
        host = p["host"]
        t = host.numpy().astype("int32", copy=True)
        inc = self.L.siftmi_gather_plan_resolve(C.byref(self.plan), t.ctypes.data, int(t.shape[0]), int(p["sent"][0]), int(p["sent"][1]))
        self._capi.check(min(inc, 0))
        if int(host[:, 2].max()) != 0:
            self.overflow_steps.append(p["step"])
        if inc == 1:                                     # every rank sees the same totals and takes the same decision: collective
            mk = min(max(int(host[:, 0].max()), 1), int(self.plan.kp_capacity))
            md = min(max(int(host[:, 1].max()), 1), int(self.plan.desc_capacity))
            _, all_kp, all_desc = _payload_gathers(p["kp"], p["desc"], p["counts"], mk, md, self.group)
            p["result"].update({"keypoints": all_kp, "descriptors": all_desc, "records_per_rank": (mk, md), "complete": True})
            self.regathered_steps.append(p["step"])
        else:
            p["result"]["complete"] = True

    def gather(self, kp_bytes: torch.Tensor, desc_bytes: torch.Tensor, counts: torch.Tensor, totals: torch.Tensor) -> Dict[str, object]:
        """The buffers of a step must stay intact until the next gather() / finish() has returned (a cut-short step is
        gathered again from them) -- the frame stream's rotating result sets give exactly that."""
        world = dist.get_world_size(self.group)
        all_totals = torch.empty((world, TOTALS), dtype=torch.int32, device=totals.device)
        dist.all_gather_into_tensor(all_totals, _totals_row(totals), group=self.group)
        self._resolve_pending()
        first = self.plan.send_kp < 0
        if first:                                        # first step: nothing to size from -> this step's own totals (one synchronous read)
            host = all_totals.cpu()
            n_kp = min(max(int(host[:, 0].max()), 1), int(self.plan.kp_capacity))
            n_desc = min(max(int(host[:, 1].max()), 1), int(self.plan.desc_capacity))
        else:
            n_kp, n_desc = int(self.plan.send_kp), int(self.plan.send_desc)
        all_counts, all_kp, all_desc = _payload_gathers(kp_bytes, desc_bytes, counts, n_kp, n_desc, self.group)
        if all_totals.is_cuda:
            host = torch.empty((world, TOTALS), dtype=torch.int32, pin_memory=True)
            host.copy_(all_totals, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = all_totals.clone(), None
        result = {"totals_device": all_totals, "counts": all_counts, "keypoints": all_kp, "descriptors": all_desc,
                  "records_per_rank": (n_kp, n_desc), "complete": False, "step": self.step}
        self._pending = {"step": self.step, "host": host, "event": ev, "sent": (n_kp, n_desc), "kp": kp_bytes, "desc": desc_bytes,
                         "counts": counts, "result": result}
        self.step += 1
        return result

    def finish(self):
        """Collective, end of stream: resolves (and if needed re-gathers) the last step; returns (regathered_steps, overflow_steps)."""
        self._resolve_pending()
        return list(self.regathered_steps), list(self.overflow_steps)
