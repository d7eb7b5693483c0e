"""Synthetic clips, block maps and server-side degradation for benchmarks and tests
(the reference ships no data; its only input `davis_test/bear.mp4` is absent).

Host numpy code, run once before any timed region.  The degradation follows the semantics of
`filter_frame_downsample` (elvis.py:2141-2169: per block, area-downscale by 2**level then bilinear
back to the block size) restated without cv2 - rounding of the bilinear step is therefore
"parity unpinned"; it only defines the input distribution, not a result that is compared.
"""
from __future__ import annotations

import numpy as np

CLIP_SEED, MAP_SEED = 20260501, 20260502


def synth_clip(seed: int, frames: int, h: int, w: int) -> np.ndarray:
    """[F,H,W,3] uint8 RGB: smooth random field panning over time + gratings + sensor noise."""
    rng = np.random.default_rng(seed)
    gh, gw = h // 16 + 4, w // 16 + 4
    base = rng.random((gh, gw, 3), dtype=np.float32)
    # bilinear x16 upsample (half-pixel centres, edge clamp) via separable interpolation matrices
    def interp_matrix(n_out, n_in, scale):
        src = (np.arange(n_out, dtype=np.float64) + 0.5) / scale - 0.5
        i0 = np.floor(src).astype(np.int64)
        t = (src - i0).astype(np.float32)
        m = np.zeros((n_out, n_in), np.float32)
        a, b = np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1)
        m[np.arange(n_out), a] += 1 - t
        m[np.arange(n_out), b] += t
        return m
    uy, ux = interp_matrix(gh * 16, gh, 16), interp_matrix(gw * 16, gw, 16)
    field = np.einsum("yi,ijc,xj->yxc", uy, base, ux, optimize=True)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    grat = np.zeros((h, w), np.float32)
    for period in (6.0, 11.0, 23.0):
        ang, ph = rng.random() * np.pi, rng.random() * 2 * np.pi
        grat += 0.08 * np.sin(2 * np.pi * (np.cos(ang) * xx + np.sin(ang) * yy) / period + ph)
    out = np.empty((frames, h, w, 3), np.uint8)
    for t in range(frames):
        oy, ox = (2 * t) % 16 + 8, (3 * t) % 16 + 8
        fr = field[oy:oy + h, ox:ox + w] * 0.8 + 0.1 + grat[:, :, None]
        fr = fr + rng.normal(0.0, 1.0 / 255.0, fr.shape).astype(np.float32)
        out[t] = np.round(np.clip(fr, 0, 1) * 255).astype(np.uint8)
    return out


def synth_level_maps(seed: int, frames: int, by: int, bx: int, probs=(0.35, 0.25, 0.25, 0.15)) -> np.ndarray:
    """[F,By,Bx] uint8 level maps: i.i.d. draw then a 5x5 majority filter so levels form regions."""
    rng = np.random.default_rng(seed)
    nl = len(probs)
    raw = rng.choice(nl, size=(frames, by, bx), p=np.asarray(probs) / np.sum(probs))
    pad = np.pad(raw, ((0, 0), (2, 2), (2, 2)), mode="edge")
    counts = np.zeros((nl, frames, by, bx), np.int32)
    for dy in range(5):
        for dx in range(5):
            win = pad[:, dy:dy + by, dx:dx + bx]
            for l in range(nl):
                counts[l] += (win == l)
    return counts.argmax(0).astype(np.uint8)


def degrade_downsample(frames: np.ndarray, levels: np.ndarray, block: int) -> np.ndarray:
    """Per block: box-mean downscale by 2**level, bilinear back to block x block (elvis.py:2154-2164)."""
    f, h, w, c = frames.shape
    by, bx = h // block, w // block
    out = frames.copy()
    blocks = frames[:, :by * block, :bx * block].reshape(f, by, block, bx, block, c).transpose(0, 1, 3, 2, 4, 5)
    res = blocks.copy()
    for lv in np.unique(levels):
        lv = int(lv)
        if lv == 0:
            continue
        fac = 2 ** lv
        s = max(1, block // fac)
        k = block // s
        sel = levels == lv
        bl = blocks[sel].astype(np.float32)                       # [N,b,b,c]
        small = np.rint(bl.reshape(-1, s, k, s, k, c).mean(axis=(2, 4)))
        src = (np.arange(block, dtype=np.float64) + 0.5) * s / block - 0.5
        i0 = np.floor(src).astype(np.int64)
        t = (src - i0).astype(np.float32)
        U = np.zeros((block, s), np.float32)
        U[np.arange(block), np.clip(i0, 0, s - 1)] += 1 - t
        U[np.arange(block), np.clip(i0 + 1, 0, s - 1)] += t
        up = np.einsum("yi,nijc,xj->nyxc", U, small, U, optimize=True)
        res[sel] = np.clip(np.rint(up), 0, 255).astype(np.uint8)
    out[:, :by * block, :bx * block] = res.transpose(0, 1, 3, 2, 4, 5).reshape(f, by * block, bx * block, c)
    return out


def make_downsample_case(frames: int, h: int, w: int, block: int = 8, max_level: int = 2,
                         clip_seed: int = CLIP_SEED, map_seed: int = MAP_SEED):
    """(clean, degraded, levels) for the Downsample/SinSR configs.  Levels are clamped to
    `max_level` (2 -> factors {1,2,4}: the single native 4x call; 3 adds the /8 level)."""
    clean = synth_clip(clip_seed, frames, h, w)
    levels = np.minimum(synth_level_maps(map_seed, frames, h // block, w // block), max_level).astype(np.uint8)
    return clean, degrade_downsample(clean, levels, block), levels
