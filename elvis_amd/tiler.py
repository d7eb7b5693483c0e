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


def adaptive_restore(restore_fn: Callable, frames: List[np.ndarray], degradation_maps: Optional[np.ndarray] = None,
                     block_size: int = 16, tile_coords: Optional[Tuple[int, int, int, int, int, int]] = None,
                     threshold: float = 0.0, **kwargs) -> List[np.ndarray]:
    """Skip `restore_fn` for tiles whose degradation-map slice never exceeds `threshold`
    (utils.py:329-394).  Integer compare on a tiny host array: host logic."""
    if degradation_maps is None:
        return restore_fn(frames=frames, **kwargs)
    should = False
    if tile_coords:
        t0, t1, y0, y1, x0, x1 = tile_coords
        by0 = y0 // block_size
        by1 = (y1 + block_size - 1) // block_size + 1
        bx0 = x0 // block_size
        bx1 = (x1 + block_size - 1) // block_size + 1
        hb, wb = degradation_maps.shape[1:]
        by1, bx1 = min(by1, hb), min(bx1, wb)
        nm = len(degradation_maps)
        tm0, tm1 = min(t0, nm), min(t1, nm)
        if tm0 < tm1:
            sl = degradation_maps[tm0:tm1, by0:by1, bx0:bx1]
            if sl.size > 0 and np.max(sl) > threshold:
                should = True
    else:
        should = True
    return restore_fn(frames=frames, **kwargs) if should else frames


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
    maps = np.stack([np.asarray(d) for d in degradation_maps])
    if maps.shape[1:] != (by, bx):
        raise ValueError(f"degradation maps {maps.shape[1:]} do not match the block grid {(by, bx)}")
    o = frames_to_device(frames, dev)
    r = frames_to_device(restored, dev)
    m = maps_to_device((maps > 0).astype(np.int32), len(frames), dev)
    return frames_to_host(ops.blend_u8(o, r, m, block_size, alpha))
