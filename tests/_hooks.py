"""Picklable stand-in restorers for the host-only tests of the directory drivers (spawned workers import
this module by name)."""
import numpy as np


def plus_device_tag(frames, maps, block_size, device, first_frame_index, **kw):
    """Adds 1 to every frame whose map has a positive entry and stamps the global frame index in pixel (0,0)."""
    out = []
    for i, f in enumerate(frames):
        g = f.copy()
        if np.max(maps[i]) > 0:
            g = (g.astype(np.int32) + 1).clip(0, 255).astype(np.uint8)
        g[0, 0, 0] = (first_frame_index + i) % 256
        out.append(g)
    return out


def temporal_window_max(frames, maps, block_size, device, first_frame_index, **kw):
    """Every output frame = max over the frames within +-1 of it that the worker was given (needs the halo
    frames to be the DECODED ones, whatever the device count)."""
    n = len(frames)
    return [np.max(np.stack(frames[max(0, i - 1):min(n, i + 2)]), axis=0) for i in range(n)]


def failing(frames, maps, block_size, device, first_frame_index, **kw):
    raise RuntimeError("boom")
