"""utils.py call surface: tiled / chunked restore wrapper, degradation gate, block-mask blend.

Same names, arguments and error behaviour as utils.py:176-394 and utils.py:1575-1601.  The task
grid and the gate are host logic; the feathered fp32 accumulate, the normalise+truncate and the
blend run as HIP kernels that reproduce numpy's float32 evaluation order bit for bit.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from . import ops
from .recompose import frames_to_device, frames_to_host, maps_to_device


def _accum_device(device) -> torch.device:
    dev = torch.device("cuda" if device in (None, "cuda") else device)
    if dev.type != "cuda":
        dev = torch.device("cuda:0")  # restore_fn may run anywhere; the blend runs on the GPU
    L.require_gpu(dev)
    return dev


def resource_aware_restore(restore_fn: Callable, frames: List[np.ndarray], tile_size: int = 512, halo: int = 16,
                           chunk_size: int = 8, chunk_overlap: int = 2, max_workers: int = 1, device: str = "cuda",
                           **kwargs) -> List[np.ndarray]:
    """Spatial tiling (step tile_size-halo) x temporal chunking (step chunk_size-chunk_overlap),
    linear feather of width halo//2 on interior edges, (i+1)/(overlap+1) temporal ramps,
    fp32 accumulate / normalise / clip / TRUNCATING uint8 cast (utils.py:176-326)."""
    if not frames:
        return []
    h, w = frames[0].shape[:2]
    n_frames = len(frames)
    do_tiling = tile_size > 0 and (h > tile_size or w > tile_size)
    do_chunking = chunk_size > 0 and n_frames > chunk_size
    if not do_tiling and not do_chunking:
        return restore_fn(frames=frames, device=device, **kwargs)

    dev = _accum_device(device)
    c = frames[0].shape[2]
    acc = torch.zeros((n_frames, h, w, c), dtype=torch.float32, device=dev)
    wsum = torch.zeros((n_frames, h, w), dtype=torch.float32, device=dev)

    if do_tiling:
        y_steps, x_steps = range(0, h, tile_size - halo), range(0, w, tile_size - halo)
    else:
        y_steps, x_steps, tile_size = [0], [0], max(h, w)
    if do_chunking:
        t_steps = range(0, n_frames, chunk_size - chunk_overlap)
    else:
        t_steps, chunk_size = [0], n_frames

    def process_task(task):
        t0, y0, x0 = task
        t1, y1, x1 = min(t0 + chunk_size, n_frames), min(y0 + tile_size, h), min(x0 + tile_size, w)
        chunk = [f[y0:y1, x0:x1] for f in frames[t0:t1]]
        try:
            out = restore_fn(frames=chunk, device=device, tile_coords=(t0, t1, y0, y1, x0, x1), **kwargs)
        except Exception as e:  # same identity fallback as utils.py:251-254
            print(f"Error processing chunk t={t0}:{t1}, y={y0}:{y1}, x={x0}:{x1}: {e}")
            out = chunk
        return (t0, t1, y0, y1, x0, x1, out)

    tasks = [(t, y, x) for t in t_steps for y in y_steps for x in x_steps]
    if max_workers > 1:
        with ThreadPoolExecutor(max_workers=max_workers) as ex:
            results = list(ex.map(process_task, tasks))
    else:
        results = [process_task(t) for t in tasks]

    for (t0, t1, y0, y1, x0, x1, out) in results:
        ch, cw = out[0].shape[:2]
        # ramps exactly as numpy builds them: a float32 array multiplied in place by float64
        # linspace ramps (top, bottom, then left, right); a tile thinner than halo//2 raises the
        # same broadcast ValueError as the reference (utils.py:287-294).
        wy = np.ones((ch,), np.float32)
        wx1 = np.ones((cw,), np.float64)
        wx2 = np.ones((cw,), np.float64)
        if do_tiling:
            fe = halo // 2
            if fe > 0:
                if y0 > 0:
                    wy[:fe] *= np.linspace(0, 1, fe)
                if y1 < h:
                    wy[-fe:] *= np.linspace(1, 0, fe)
                if x0 > 0:
                    wx1[:fe] *= np.linspace(0, 1, fe)
                if x1 < w:
                    wx2[-fe:] *= np.linspace(1, 0, fe)
        tile_d = frames_to_device([np.ascontiguousarray(o) for o in out], dev)
        wy_d = torch.from_numpy(wy).to(dev)
        wx1_d = torch.from_numpy(wx1).to(dev)
        wx2_d = torch.from_numpy(wx2).to(dev)
        for i in range(len(out)):
            gt = t0 + i
            tw = 1.0
            if do_chunking:
                if t0 > 0 and i < chunk_overlap:
                    tw *= (i + 1) / (chunk_overlap + 1)
                if t1 < n_frames and i >= (len(out) - chunk_overlap):
                    tw *= (len(out) - i) / (chunk_overlap + 1)
            ops.tile_accumulate(acc[gt], wsum[gt], tile_d[i], wy_d, wx1_d, wx2_d, y0, x0, float(np.float32(tw)))

    final = torch.empty((n_frames, h, w, c), dtype=torch.uint8, device=dev)
    for i in range(n_frames):
        ops.tile_normalize(acc[i], wsum[i], out=final[i])
    return frames_to_host(final)


def _block_span(lo: int, hi: int, block: int, limit: int) -> slice:
    """Blocks that pixel range [lo, hi) touches, plus the one guard block after it that the reference's
    gate also inspects, clipped to the map."""
    return slice(lo // block, min(-(-hi // block) + 1, limit))


def adaptive_restore(restore_fn: Callable, frames: List[np.ndarray], degradation_maps: Optional[np.ndarray] = None,
                     block_size: int = 16, tile_coords: Optional[Tuple[int, int, int, int, int, int]] = None,
                     threshold: float = 0.0, **kwargs) -> List[np.ndarray]:
    """Degradation gate for the tiler (interface of utils.py:329-394, meant to be `functools.partial`-ed
    around a `restore_fn`): a tile whose window of the FULL `(N, By, Bx)` map holds no value above
    `threshold` is returned untouched; every other call goes to `restore_fn(frames=frames, **kwargs)`.
    Without maps or without `tile_coords` there is nothing to gate on and the restorer always runs.
    An integer compare on a few map entries: host logic (behaviour pinned by tests/golden/gate.npz)."""
    if degradation_maps is not None and tile_coords:
        maps = np.asarray(degradation_maps)
        t0, t1, y0, y1, x0, x1 = tile_coords
        window = maps[min(t0, len(maps)):min(t1, len(maps)),
                      _block_span(y0, y1, block_size, maps.shape[1]),
                      _block_span(x0, x1, block_size, maps.shape[2])]
        if not (window > threshold).any():
            return frames
    return restore_fn(frames=frames, **kwargs)


def _extract_tile_with_halo(frame: np.ndarray, y: int, x: int, tile_h: int, tile_w: int, halo: int):
    """Tile (y, x, tile_h, tile_w) of `frame` grown by `halo` pixels on every side where the frame allows,
    as a copy, with the (top, left, bottom, right) bounds that crop the halo off a same-size result
    (interface of utils.py:1227-1250; pinned by tests/golden/tiler.npz).  Index math on the host."""
    h, w = frame.shape[:2]
    top, left = min(halo, max(y, 0)), min(halo, max(x, 0))
    tile = frame[y - top:min(h, y + tile_h + halo), x - left:min(w, x + tile_w + halo)].copy()
    return tile, (top, left, top + tile_h, left + tile_w)


extract_tile_with_halo = _extract_tile_with_halo


def _nearest_rows(src_n: int, dst_n: int) -> np.ndarray:
    """INTER_NEAREST source index of every destination index: floor(dst * src_n / dst_n)."""
    return (np.arange(dst_n, dtype=np.int64) * src_n) // dst_n


def blended_restoration(frames, degradation_maps, block_size, alpha=1.0, restore_fn=None, device="cuda", **kwargs):
    """utils.py:1575-1601: tiled restore, then out = orig*(1-a*m) + restored*(a*m), m=(map>0)."""
    if restore_fn is None:
        from .restore import restore_with_sinsr_naive
        restore_fn = restore_with_sinsr_naive
    restored = resource_aware_restore(restore_fn, frames, device=device, **kwargs)
    if not frames:
        return []
    dev = _accum_device(device)
    h, w = frames[0].shape[:2]
    by, bx = h // block_size, w // block_size
    maps = np.stack([np.asarray(d) for d in degradation_maps])[: len(frames)]
    if len(maps) != len(frames):
        raise ValueError(f"{len(maps)} degradation maps for {len(frames)} frames")
    if maps.shape[1:] != (by, bx):   # the reference NEAREST-resizes a map of another shape to the grid (utils.py:1587)
        maps = maps[:, _nearest_rows(maps.shape[1], by)][:, :, _nearest_rows(maps.shape[2], bx)]
    o = frames_to_device(frames, dev)
    r = frames_to_device(restored, dev)
    m = maps_to_device((maps > 0).astype(np.int32), len(frames), dev)
    return frames_to_host(ops.blend_u8(o, r, m, block_size, alpha))
