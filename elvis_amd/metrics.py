"""On-device quality metrics of the parity / quality report: whole-frame and masked MSE / PSNR
(integer-exact sums of squared differences on the device, the reference's formulas on the host) and per-block SSIM.

  calculate_mse / calculate_psnr   presley.py:226-245  (PSNR = 10 log10(range^2 / mse), inf at mse == 0)
  masked_mse / masked_psnr         elvis.py:627-671    (PSNR = 20 log10(255 / sqrt(mse)) capped at 100 dB)
  calculate_block_ssim             utils.py:572-608    (pytorch_msssim.ssim per block; restated, package absent)
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from . import ops
from ._lib import check, lib, ptr
from .recompose import frames_to_device


def _dev(device) -> torch.device:
    dev = torch.device("cuda:0" if str(device) == "cuda" else device)
    L.require_gpu(dev)
    return dev


def _sse(reference_frames, distorted_frames, device, masks=None):
    dev = _dev(device)
    with torch.cuda.device(dev):
        a, b = frames_to_device(list(reference_frames), dev), frames_to_device(list(distorted_frames), dev)
        if a.shape != b.shape:
            raise ValueError("frame sequences differ in shape")
        m = None
        if masks is not None:
            m = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(k).astype(bool) for k in masks]).astype(np.uint8))).to(dev)
        sse, cnt = ops.sse_u8(a, b, m)
        return sse.cpu().numpy(), cnt.cpu().numpy()


def calculate_mse(reference_frames: Sequence[np.ndarray], distorted_frames: Sequence[np.ndarray], device="cuda:0") -> List[float]:
    """Per-frame MSE (presley.py:226-232)."""
    if not len(reference_frames):
        return []
    sse, cnt = _sse(reference_frames, distorted_frames, device)
    return [float(s) / float(c) for s, c in zip(sse, cnt)]


def calculate_psnr(reference_frames, distorted_frames, data_range: float = 255.0, device="cuda:0") -> List[float]:
    """Per-frame PSNR, 10 log10(data_range^2 / mse), inf for identical frames (presley.py:235-245)."""
    return [float("inf") if m == 0 else 10.0 * math.log10(data_range ** 2 / m)
            for m in calculate_mse(reference_frames, distorted_frames, device)]


def masked_mse(ref: np.ndarray, dec: np.ndarray, mask: Optional[np.ndarray] = None, device="cuda:0") -> float:
    """elvis.py:653-671: MSE over the masked pixels (all channels), 0.0 for an empty mask."""
    sse, cnt = _sse([ref], [dec], device, None if mask is None else [mask])
    return 0.0 if cnt[0] == 0 else float(sse[0]) / float(cnt[0])


def masked_psnr(ref: np.ndarray, dec: np.ndarray, mask: Optional[np.ndarray] = None, device="cuda:0") -> float:
    """elvis.py:627-650: 20 log10(255 / sqrt(mse)) capped at 100 dB; 100 for an empty mask or mse < 1e-10."""
    if mask is not None and not np.any(np.asarray(mask).astype(bool)):
        return 100.0
    mse = masked_mse(ref, dec, mask, device)
    return 100.0 if mse < 1e-10 else float(min(20.0 * math.log10(255.0 / math.sqrt(mse)), 100.0))


def ssim_window(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    coords = np.arange(size, dtype=np.float32) - size // 2
    g = np.exp(-(coords ** 2) / np.float32(2 * sigma ** 2)).astype(np.float32)
    return (g / g.sum()).astype(np.float32)


def block_ssim_device(a: torch.Tensor, b: torch.Tensor, block_size: int) -> torch.Tensor:
    """[n,H,W,C] uint8 x2 on the device -> float32 [n, H // b, W // b]."""
    ops._chk_u8(a, b)
    if a.shape != b.shape:
        raise ValueError("frame tensors differ in shape")
    n, h, w, c = a.shape
    out = torch.empty((n, h // block_size, w // block_size), dtype=torch.float32, device=a.device)
    win = torch.from_numpy(ssim_window()).to(a.device)
    check(lib().elvis_block_ssim_u8(ptr(a), ptr(b), ptr(out), ptr(win), n, h, w, c, block_size, ops._s(a)), a.device)
    return out


def calculate_block_ssim(frames1: Sequence[np.ndarray], frames2: Sequence[np.ndarray], block_size: int,
                         device="cuda:0") -> List[np.ndarray]:
    """Per-block SSIM maps (utils.py:572-608), one (H // b, W // b) float32 array per frame pair."""
    if not len(frames1):
        return []
    dev = _dev(device)
    with torch.cuda.device(dev):
        m = block_ssim_device(frames_to_device(list(frames1), dev), frames_to_device(list(frames2), dev), block_size).cpu().numpy()
    return [m[i] for i in range(m.shape[0])]
