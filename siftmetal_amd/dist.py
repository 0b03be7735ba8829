"""Frame-per-GPU sharding and the descriptor all-gather (the only exchange step of the path).

Frames are independent units (the reference keeps no cross-frame state: SIFT.swift holds only
scratch), so a stream of frames shards with no data-path collective; after a batch every rank
publishes its packed keypoint / descriptor buffers with ONE count exchange + padded all-gathers
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
The functions here are device-agnostic tensor plumbing; no SIFT arithmetic.
"""
from typing import Dict, List

import torch
import torch.distributed as dist

KP_BYTES, DESC_BYTES = 44, 136


def shard_frames(n_frames: int, world_size: int, rank: int) -> List[int]:
    """frame i -> rank i mod world_size; returns this rank's frame indices."""
    return list(range(rank, n_frames, world_size))


def gather_results(kp_bytes: torch.Tensor, desc_bytes: torch.Tensor, counts: torch.Tensor, totals: torch.Tensor,
                   group=None) -> Dict[str, object]:
    """All-gather one batch's packed results.

    kp_bytes   uint8 [kp_capacity * 44]    packed siftmi_keypoint records (first totals[0] valid)
    desc_bytes uint8 [desc_capacity * 136] packed siftmi_descriptor records (first totals[1] valid)
    counts     int32 [2, n_frames, n_octaves]
    totals     int32 [2] = {n_keypoints, n_descriptors}
    Returns per-rank views: {"totals": [world,2] (host), "counts": [world,2,F,O], "keypoints": [world, max_kp*44],
    "descriptors": [world, max_desc*136]} -- rank r's valid bytes are the first totals[r]*record_size of row r.
    """
    world = dist.get_world_size(group)
    all_totals = torch.empty((world, 2), dtype=torch.int32, device=totals.device)
    dist.all_gather_into_tensor(all_totals, totals.reshape(1, 2).contiguous(), group=group)
    host_totals = all_totals.cpu()                       # the one host sync of the exchange
    max_kp = max(int(host_totals[:, 0].max()), 1)
    max_desc = max(int(host_totals[:, 1].max()), 1)
    if max_kp * KP_BYTES > kp_bytes.numel() or max_desc * DESC_BYTES > desc_bytes.numel():
        raise RuntimeError("gather_results: a peer holds more records than this rank's buffer capacity")
    all_counts = torch.empty((world,) + tuple(counts.shape), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(all_counts, counts.contiguous().unsqueeze(0), group=group)
    all_kp = torch.empty((world, max_kp * KP_BYTES), dtype=torch.uint8, device=kp_bytes.device)
    dist.all_gather_into_tensor(all_kp, kp_bytes[:max_kp * KP_BYTES].unsqueeze(0), group=group)
    all_desc = torch.empty((world, max_desc * DESC_BYTES), dtype=torch.uint8, device=desc_bytes.device)
    dist.all_gather_into_tensor(all_desc, desc_bytes[:max_desc * DESC_BYTES].unsqueeze(0), group=group)
    return {"totals": host_totals, "counts": all_counts, "keypoints": all_kp, "descriptors": all_desc}
