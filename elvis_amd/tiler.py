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


def _windows(total: int, size: int, step: int) -> List[Tuple[int, int]]:
    """[start, end) windows of `size` samples every `step` samples over [0, total), the last ones clipped."""
    return [(s0, min(s0 + size, total)) for s0 in range(0, total, step)]


def _edge_ramp(n: int, width: int, fade_in: bool, fade_out: bool, dtype) -> np.ndarray:
    """Per-sample weights of a window of n samples: 1 inside, a linear 0->1 ramp over the first `width` samples
    when the window has a neighbour before it, 1->0 over the last `width` when it has one after it.  Built the
    way numpy evaluates the reference's in-place products (a `dtype` array of ones multiplied by float64
    linspace ramps), so the device accumulate can reproduce its rounding; a window thinner than `width` raises
    the same broadcast ValueError as the reference (utils.py:287-294)."""
    wgt = np.ones((n,), dtype)
    if width > 0:
        if fade_in:
            wgt[:width] *= np.linspace(0, 1, width)
        if fade_out:
            wgt[-width:] *= np.linspace(1, 0, width)
    return wgt


def _temporal_weight(i: int, count: int, overlap: int, has_prev: bool, has_next: bool) -> float:
    """Weight of frame i of a temporal chunk of `count` frames: (i+1)/(overlap+1) over the first `overlap` frames
    of a chunk that has a predecessor, mirrored at the end of one that has a successor (utils.py:300-310)."""
    wgt = 1.0
    if has_prev and i < overlap:
        wgt *= (i + 1) / (overlap + 1)
    if has_next and i >= count - overlap:
        wgt *= (count - i) / (overlap + 1)
    return wgt


def resource_aware_restore(restore_fn: Callable, frames: List[np.ndarray], tile_size: int = 512, halo: int = 16,
                           chunk_size: int = 8, chunk_overlap: int = 2, max_workers: int = 1, device: str = "cuda",
                           **kwargs) -> List[np.ndarray]:
    """The tiler every `utils.py` restorer goes through (interface and results of utils.py:176-326, pinned by
    tests/golden/tiler.npz): spatial tiles of `tile_size` every `tile_size - halo` pixels x temporal chunks of
    `chunk_size` every `chunk_size - chunk_overlap` frames; each task calls
    `restore_fn(frames=tile_stack, device=, tile_coords=(t0, t1, y0, y1, x0, x1), **kwargs)`; results are blended
    with linear feathers of width halo // 2 on interior tile edges and temporal ramps, accumulated in fp32,
    normalised, clipped and TRUNCATED to uint8.  No tiling and no chunking needed -> `restore_fn` is called once,
    directly.  A task whose `restore_fn` raises is replaced by its input (the reference's behaviour, with a
    message) - a restorer must not rely on that.  The accumulate / normalise run as HIP kernels."""
    if not frames:
        return []
    h, w = frames[0].shape[:2]
    n_frames = len(frames)
    tiled = tile_size > 0 and (h > tile_size or w > tile_size)
    chunked = chunk_size > 0 and n_frames > chunk_size
    if not (tiled or chunked):
        return restore_fn(frames=frames, device=device, **kwargs)

    dev = _accum_device(device)
    c = frames[0].shape[2]
    acc = torch.zeros((n_frames, h, w, c), dtype=torch.float32, device=dev)
    wsum = torch.zeros((n_frames, h, w), dtype=torch.float32, device=dev)

    side = tile_size if tiled else max(h, w)
    rows = _windows(h, side, tile_size - halo) if tiled else [(0, h)]
    cols = _windows(w, side, tile_size - halo) if tiled else [(0, w)]
    spans = _windows(n_frames, chunk_size, chunk_size - chunk_overlap) if chunked else [(0, n_frames)]
    tasks = [(t, y, x) for t in spans for y in rows for x in cols]

    def run(task):
        (t0, t1), (y0, y1), (x0, x1) = task
        stack = [f[y0:y1, x0:x1] for f in frames[t0:t1]]
        try:
            return restore_fn(frames=stack, device=device, tile_coords=(t0, t1, y0, y1, x0, x1), **kwargs)
        except Exception as exc:
            print(f"restore_fn failed on task t={t0}:{t1}, y={y0}:{y1}, x={x0}:{x1} ({exc}); passing the input through")
            return stack

    if max_workers > 1:
        with ThreadPoolExecutor(max_workers=max_workers) as pool:
            outputs = list(pool.map(run, tasks))
    else:
        outputs = [run(t) for t in tasks]

    feather = halo // 2 if tiled else 0
    for ((t0, t1), (y0, y1), (x0, x1)), out in zip(tasks, outputs):
        th, tw = out[0].shape[:2]
        # the reference multiplies ONE float32 weight image by the y ramps and then by the left and right x ramps;
        # keeping the two x ramps apart (float64) lets the kernel apply the products in the same order
        wy = _edge_ramp(th, feather, y0 > 0, y1 < h, np.float32)
        wx_left = _edge_ramp(tw, feather, x0 > 0, False, np.float64)
        wx_right = _edge_ramp(tw, feather, False, x1 < w, np.float64)
        tile_d = frames_to_device([np.ascontiguousarray(o) for o in out], dev)
        wy_d, wl_d, wr_d = (torch.from_numpy(a).to(dev) for a in (wy, wx_left, wx_right))
        for i in range(len(out)):
            tw_i = _temporal_weight(i, len(out), chunk_overlap, t0 > 0, t1 < n_frames) if chunked else 1.0
            ops.tile_accumulate(acc[t0 + i], wsum[t0 + i], tile_d[i], wy_d, wl_d, wr_d, y0, x0, float(np.float32(tw_i)))

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
