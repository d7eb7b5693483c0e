"""Frame-sharded multi-GPU restoration: one process (rank) per GPU, contiguous frame ranges by the
`chunk_for_devices` rule (elvis.py:255-280; same rule as `_split_ranges`, elvis.py:3046-3060), and ONE
all-gather of the restored uint8 frames to reassemble the decoded sequence (RCCL over xGMI on the
MI355X node; `gloo` on CPU for the ordering tests).

The reference's own "gather" is the filesystem / python lists reassembled by chunk_id
(elvis.py:348-353, 2983-2985); there is no collective to translate.  Frames are independent, so
there is no data-path exchange other than this final gather.  Sampler noise is keyed on the GLOBAL
frame index, so the result does not depend on the world size (unlike elvis.py:3127).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .sharding import rank_frame_range


def shard_sizes(total: int, world_size: int) -> List[int]:
    return [rank_frame_range(total, world_size, r)[1] - rank_frame_range(total, world_size, r)[0]
            for r in range(world_size)]


def all_gather_frames(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """Gather per-rank shards `[n_r, H, W, C]` uint8 (n_r by the chunk_for_devices rule) into the
    full `[total, H, W, C]` sequence on every rank, in frame order.

    One `all_gather_into_tensor` on a contiguous buffer; shards are padded to the largest shard
    when `total % world != 0` (30 frames / 8 ranks -> 4,4,4,4,4,4,3,3) and trimmed afterwards.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} frames, expected {sizes[rank]}")
    if world == 1:
        return local
    nmax = max(sizes)
    frame_shape = tuple(local.shape[1:])
    if local.shape[0] == nmax and local.is_contiguous():
        send = local
    else:
        send = torch.zeros((nmax,) + frame_shape, dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    recv = torch.empty((world * nmax,) + frame_shape, dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "gloo" and not hasattr(dist, "_all_gather_base"):
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send, group=group)
        recv = torch.cat(parts, 0)
    else:
        try:
            dist.all_gather_into_tensor(recv, send, group=group)
        except (RuntimeError, NotImplementedError):
            parts = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(parts, send, group=group)
            recv = torch.cat(parts, 0)
    if all(s == nmax for s in sizes):
        return recv
    recv = recv.view((world, nmax) + frame_shape)
    return torch.cat([recv[r, : sizes[r]] for r in range(world)], 0)


def restore_clip_sharded(frames: Sequence[np.ndarray], maps: np.ndarray,
                         restore_shard: Callable[[List[np.ndarray], np.ndarray, int], List[np.ndarray]],
                         device: Optional[torch.device] = None, group=None) -> List[np.ndarray]:
    """Every rank receives the whole decoded clip (host memory, read-only), restores its own
    contiguous range with `restore_shard(frames, maps, first_global_index)` and all ranks get the
    full restored sequence back, in order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    total = len(frames)
    s, e = rank_frame_range(total, world, rank)
    out = restore_shard(list(frames[s:e]), np.asarray(maps)[s:e], s) if e > s else []
    if world == 1:
        return out
    dev = device if device is not None else torch.device("cpu")
    h, w, c = frames[0].shape
    local = torch.from_numpy(np.stack(out)).to(dev) if out else torch.empty((0, h, w, c), dtype=torch.uint8, device=dev)
    full = all_gather_frames(local, total, group).cpu().numpy()
    return [full[i] for i in range(total)]
