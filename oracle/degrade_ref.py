"""CPU oracle (numpy) of the server-side per-block degrade filters.

TEST INFRASTRUCTURE - NOT PRODUCT CODE (see oracle/glue_ref.py header for the rule).

`filter_frame_downsample` / `filter_frame_gaussian` follow elvis.py:2141-2196 block for block; their cv2 calls
(`resize` INTER_AREA / INTER_LINEAR on uint8, `GaussianBlur` 5x5 sigma 1 with the default BORDER_REFLECT_101)
cannot run here (no cv2), so they are restated from OpenCV's documented uint8 arithmetic:
INTER_AREA as in glue_ref.area_downscale_u8; INTER_LINEAR with 11-bit coefficients, horizontal pass in int32,
vertical pass `((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`; the Gaussian as two float32 separable passes
with `getGaussianKernel(5, 1)` taps and a round-half-even uint8 cast per call.  PARITY UNPINNED against cv2
itself; the device kernels are pinned bit-exactly against THIS file.  `dct_dampen` is the build's own
definition of the DCT degrade (the reference has no code for it)."""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np


def gaussian_taps() -> Tuple[np.float32, np.float32, np.float32]:
    """cv2.getGaussianKernel(5, 1.0) (float32 after normalisation in float64): (k0, k1, k2) of (k0 k1 k2 k1 k0)."""
    k = np.exp(-np.arange(-2, 3, dtype=np.float64) ** 2 / 2.0)
    k = (k / k.sum()).astype(np.float32)
    return k[0], k[1], k[2]


def _linear_coef(d: int, s: int, b: int):
    f = np.float32((d + 0.5) * (s / b) - 0.5)
    i = int(math.floor(f))
    f = np.float32(f - np.float32(i))
    if i < 0:
        i, f = 0, np.float32(0)
    if i >= s - 1:
        i, f = s - 1, np.float32(0)
    a0 = int(np.rint(np.float32(np.float32(1.0) - f) * np.float32(2048.0)))
    a1 = int(np.rint(f * np.float32(2048.0)))
    return i, a0, a1


def _downsample_block(block: np.ndarray, level: int) -> np.ndarray:
    b = block.shape[0]
    fac = 1 << min(max(level, 0), 4)
    if fac <= 1:
        return block.copy()
    s = b // fac
    if s < 1:
        s, fac = 1, b
    sums = block.reshape(s, fac, s, fac, -1).astype(np.uint32).sum(axis=(1, 3))
    if fac == 2:
        small = ((sums + 2) >> 2).astype(np.int64)
    else:
        small = np.clip(np.rint(sums.astype(np.float32) * np.float32(1.0 / (fac * fac))), 0, 255).astype(np.int64)
    coef = [_linear_coef(d, s, b) for d in range(b)]
    out = np.empty_like(block)
    for y in range(b):
        y0, b0, b1 = coef[y]
        y1 = min(y0 + 1, s - 1)
        for x in range(b):
            x0, a0, a1 = coef[x]
            x1 = min(x0 + 1, s - 1)
            r0 = small[y0, x0] * a0 + small[y0, x1] * a1
            r1 = small[y1, x0] * a0 + small[y1, x1] * a1
            v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
            out[y, x] = np.clip(v, 0, 255).astype(np.uint8)
    return out


def degrade_downsample(frame: np.ndarray, levels: np.ndarray, block: int) -> np.ndarray:
    """The image half of filter_frame_downsample (elvis.py:2149-2167) for a given level map."""
    out = frame.copy()
    for by in range(levels.shape[0]):
        for bx in range(levels.shape[1]):
            if levels[by, bx] > 0:
                sl = (slice(by * block, (by + 1) * block), slice(bx * block, (bx + 1) * block))
                out[sl] = _downsample_block(frame[sl], int(levels[by, bx]))
    return out


def filter_frame_downsample(image: np.ndarray, frame_scores: np.ndarray, block_size: int):
    """elvis.py:2141-2169: scores in [0,1] -> levels = round(score * log2(block_size)); returns (image, levels)."""
    levels = np.round(frame_scores * int(np.log2(block_size))).astype(np.int32)
    return degrade_downsample(image, levels, block_size), levels


def _reflect101(i: int, n: int) -> int:
    if n == 1:
        return 0
    while i < 0 or i >= n:
        i = -i if i < 0 else 2 * (n - 1) - i
    return i


def _blur_block(block: np.ndarray, rounds: int) -> np.ndarray:
    b = block.shape[0]
    k0, k1, k2 = gaussian_taps()
    kk = [k0, k1, k2, k1, k0]
    idx = [[_reflect101(x + d - 2, b) for d in range(5)] for x in range(b)]
    cur = block.copy()
    for _ in range(rounds):
        f = cur.astype(np.float32)
        tmp = np.zeros_like(f)
        for d in range(5):                     # a = a + k[d] * v, d ascending, float32
            tmp = tmp + kk[d] * f[:, [idx[x][d] for x in range(b)]]
        acc = np.zeros_like(f)
        for d in range(5):
            acc = acc + kk[d] * tmp[[idx[y][d] for y in range(b)], :]
        cur = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    return cur


def degrade_gaussian(frame: np.ndarray, rounds: np.ndarray, block: int) -> np.ndarray:
    out = frame.copy()
    for by in range(rounds.shape[0]):
        for bx in range(rounds.shape[1]):
            if rounds[by, bx] > 0:
                sl = (slice(by * block, (by + 1) * block), slice(bx * block, (bx + 1) * block))
                out[sl] = _blur_block(frame[sl], int(rounds[by, bx]))
    return out


def filter_frame_gaussian(image: np.ndarray, frame_scores: np.ndarray, block_size: int):
    """elvis.py:2171-2196: rounds = round(score * 10); returns (image, rounds)."""
    rounds = np.round(frame_scores * 10).astype(np.int32)
    return degrade_gaussian(image, rounds, block_size), rounds


def dct_basis() -> np.ndarray:
    """basis[u][x] = C(u) cos((2x+1) u pi / 16), C(0) = sqrt(1/8), C(u>0) = sqrt(2/8); float32 of the float64 value."""
    u = np.arange(8, dtype=np.float64)[:, None]
    x = np.arange(8, dtype=np.float64)[None, :]
    b = np.cos((2 * x + 1) * u * np.pi / 16.0) * np.sqrt(2.0 / 8.0)
    b[0] *= np.sqrt(0.5)
    return b.astype(np.float32)


def dct_gain(n_levels: int) -> np.ndarray:
    """gain[level][u][v] = 2^(-level * (u + v) / 14), float32 of the float64 value."""
    uv = (np.arange(8)[:, None] + np.arange(8)[None, :]).astype(np.float64)
    return np.stack([np.exp2(-lv * uv / 14.0) for lv in range(n_levels)]).astype(np.float32)


def _mm(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """float32 matrix product with the device kernel's evaluation order: acc = acc + a[:, k] * b[k, :], k ascending."""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(a.shape[1]):
        acc = acc + a[:, k:k + 1] * b[k:k + 1, :]
    return acc


def dct_dampen(frame: np.ndarray, levels: np.ndarray, n_levels: int = 4) -> np.ndarray:
    B, G = dct_basis(), dct_gain(n_levels)
    out = frame.copy()
    for by in range(levels.shape[0]):
        for bx in range(levels.shape[1]):
            lv = int(np.clip(levels[by, bx], 0, n_levels - 1))
            if lv == 0:
                continue
            for c in range(frame.shape[2]):
                X = frame[by * 8:(by + 1) * 8, bx * 8:(bx + 1) * 8, c].astype(np.float32)
                Y = _mm(_mm(B, X), B.T.copy()) * G[lv]
                Z = _mm(_mm(B.T.copy(), Y), B)
                out[by * 8:(by + 1) * 8, bx * 8:(bx + 1) * 8, c] = np.clip(np.rint(Z), 0, 255).astype(np.uint8)
    return out
