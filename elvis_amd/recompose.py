"""Block-map-guided recompose drivers, on device (the reference's a1-a3, a6, a15, a16 rows of
SURVEY.md 8a).  Host numpy frames go to HBM once, every paste / downscale / blend runs as a
HIP kernel, and the restored frames come back once.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from . import ops


# ----------------------------------------------------------------------------- host views
def split_image_into_blocks(image: np.ndarray, block_size: int) -> np.ndarray:
    """(H,W,C) -> (By,Bx,b,b,C) view; ValueError when H or W is not divisible (elvis.py:1369-1385).
    Pure index math (no arithmetic), so it stays a host numpy view like the reference's."""
    h, w, c = image.shape
    if h % block_size != 0 or w % block_size != 0:
        raise ValueError("Image dimensions must be divisible by block_size.")
    return image.reshape(h // block_size, block_size, w // block_size, block_size, c).swapaxes(1, 2)


def combine_blocks_into_image(blocks: np.ndarray) -> np.ndarray:
    """Inverse of split_image_into_blocks (elvis.py:1429-1434)."""
    by, bx, b, _, c = blocks.shape
    return blocks.swapaxes(1, 2).reshape(by * b, bx * b, c)


def frames_to_device(frames: Sequence[np.ndarray], device) -> torch.Tensor:
    L.require_gpu(device)
    if not frames:
        raise ValueError("no frames")
    shp = frames[0].shape
    for f in frames:
        if f.shape != shp or f.dtype != np.uint8 or f.ndim != 3:
            raise ValueError("frames must be uint8 (H,W,C) arrays of one shape")
    host = torch.from_numpy(np.ascontiguousarray(np.stack(frames, axis=0)))
    return host.to(device, non_blocking=False)


def frames_to_host(t: torch.Tensor) -> List[np.ndarray]:
    arr = t.cpu().numpy()
    return [arr[i] for i in range(arr.shape[0])]


def maps_to_device(maps, n: int, device) -> torch.Tensor:
    m = np.asarray(maps)
    if m.ndim == 2:
        m = m[None]
    if m.ndim != 3 or m.shape[0] != n:
        raise ValueError(f"maps must be (frames, blocks_y, blocks_x); got {m.shape} for {n} frame(s)")
    return torch.from_numpy(np.ascontiguousarray(m.astype(np.int32))).to(device)


# ----------------------------------------------------------------------------- a3
def upscale_adaptive_device(frames_d: torch.Tensor, levels_d: torch.Tensor, block_size: int,
                            upsample_dev: Callable[[torch.Tensor], torch.Tensor], *, sr_scale: int = 2,
                            rounding: int = L.ROUND_CV2) -> torch.Tensor:
    """Staged coarse-to-fine recompose of elvis.py:2522-2600 on device, batched over frames.

    frames_d [n,H,W,3] u8, levels_d [n,By,Bx] int32 (log2 of the per-block downscale factor,
    elvis.py:2558).  `upsample_dev` maps [n,h,w,3] u8 -> [n,s*h,s*w,3] u8 with s = sr_scale.
    With sr_scale=2 this is the reference loop line by line; with sr_scale=4 stages advance by
    4x while the remaining factor allows and a trailing 2x stage area-halves the 4x output (what
    RealESRGANer.enhance(outscale=2) does with its x4 net, elvis.py:2515).

    The frames of a batch share the stage schedule: it starts from the LARGEST factor in the
    batch (for a frame whose own maximum is smaller the extra leading stages only produce
    pixels that the later paste of its own blocks overwrites... no: they change SR context),
    so callers that need exact per-frame semantics pass one frame at a time or pre-group
    frames by max level - `restore_frames_sinsr` does the latter.
    """
    n, H, W, C = frames_d.shape
    by, bx = levels_d.shape[1:]
    if H % block_size or W % block_size:
        raise ValueError("Image dimensions must be divisible by block_size.")
    if (by, bx) != (H // block_size, W // block_size):
        raise ValueError(f"map grid {(by, bx)} does not match frame/block grid {(H // block_size, W // block_size)}")
    max_level = int(levels_d.max().item())
    if max_level < 0 or (1 << max_level) > block_size:
        raise ValueError(f"downscale level {max_level} out of range for block_size {block_size}")
    f = 1 << max_level
    cur = ops.area_downscale_u8(frames_d, f, rounding) if f > 1 else frames_d.clone()
    lev = levels_d.contiguous()
    while f > 1:
        if sr_scale == 2 or f < sr_scale:
            nf = f // 2
            up = upsample_dev(cur)
            if up.shape[1] != cur.shape[1] * 2:
                up = ops.area_downscale_u8(up, up.shape[1] // (cur.shape[1] * 2), rounding)
        else:
            nf = f // sr_scale
            up = upsample_dev(cur)
        # level index of the new factor: nf = 2**lv
        lv = nf.bit_length() - 1
        ref = ops.area_downscale_u8(frames_d, nf, rounding) if nf > 1 else frames_d
        new_lev = torch.empty_like(lev)
        # factor <= nf  <=>  level <= lv : take the (area-downscaled) decoded frame, else keep SR
        cur = ops.recompose_u8(ref, up, lev, block_size // nf, lv, map_out=new_lev, clamp_to=lv)
        lev = new_lev
        f = nf
    return cur


def rounds_recompose_device(frames_d: torch.Tensor, maps_d: torch.Tensor, block_size: int,
                            restore_dev: Callable[[torch.Tensor], torch.Tensor], batch_size: int = 4,
                            max_rounds: Optional[int] = None) -> torch.Tensor:
    """The iterative round loop of _instantir_chunk_worker (elvis.py:2947-2981) on device:
    for r in range(max(map)): restore every frame that still has map>0; re-paste the ORIGINAL
    decoded blocks where the remaining level <= 0; decrement positive entries."""
    n, H, W, C = frames_d.shape
    if H % block_size or W % block_size:
        raise ValueError("Image dimensions must be divisible by block_size.")
    cur = frames_d.clone()
    m = maps_d.clone()
    rounds = int(m.max().item()) if m.numel() else 0
    if max_rounds is not None:
        rounds = min(rounds, max_rounds)
    step = max(1, batch_size)
    for _ in range(rounds):
        active = torch.nonzero((m > 0).flatten(1).any(dim=1)).flatten().tolist()
        if not active:
            break
        for off in range(0, len(active), step):
            idx = torch.tensor(active[off:off + step], device=frames_d.device)
            restored = restore_dev(cur[idx].contiguous())
            pasted = ops.recompose_u8(frames_d[idx].contiguous(), restored, m[idx].contiguous(), block_size, 0)
            cur[idx] = pasted
        m = torch.where(m > 0, m - 1, m)
    return cur


# ----------------------------------------------------------------------------- a15 / a16
def blend_by_map_device(orig_d, rest_d, maps_d, block_size: int, alpha: float) -> torch.Tensor:
    """utils.py:1581-1599 on device (mask = map>0 upsampled NEAREST; fp32 blend; truncating cast)."""
    n, H, W, C = orig_d.shape
    by, bx = maps_d.shape[1:]
    if (by, bx) != (H // block_size, W // block_size):
        raise ValueError("degradation map does not match the block grid")
    return ops.blend_u8(orig_d, rest_d, maps_d, block_size, alpha)


def restore_video_adaptively(restore_fn: Callable, frames: List[np.ndarray], degradation_maps: List[np.ndarray],
                             block_size: int = 16, device="cuda:0", **kwargs) -> List[np.ndarray]:
    """presley.py:1219-1275: run `restore_fn(frames=, degradation_level=L, **kw)` once per
    distinct level, then pick per block on device.  Pixels outside the floored block grid stay
    0 like the reference (presley.py:1265)."""
    if not frames:
        return []
    h, w = frames[0].shape[:2]
    n = len(frames)
    levels = set()
    for d in degradation_maps:
        levels.update(np.unique(d).tolist())
    levels = sorted(levels)
    versions = []
    for level in levels:
        kw = dict(kwargs)
        kw["degradation_level"] = level
        res = restore_fn(frames=frames, **kw)
        if isinstance(res, tuple) and len(res) == 2:
            res = res[0]
        versions.append(frames_to_device(res, device))
    int_levels = [int(l) for l in levels]
    if any(l < 0 for l in int_levels):
        raise ValueError("negative degradation level")
    table = np.full(max(int_levels) + 1, -1, np.int32)
    for slot, l in enumerate(int_levels):
        table[l] = slot
    maps_d = maps_to_device(np.stack([np.asarray(d) for d in degradation_maps]), n, device)
    out = ops.select_levels_u8(versions, torch.from_numpy(table).to(device), maps_d, block_size)
    return frames_to_host(out)
