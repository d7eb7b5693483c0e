"""CPU oracle (numpy) for the block-map glue of the ELVIS client-side restore path.

TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this package.  The product path
(`elvis_amd/`) never imports it and fails loudly when the HIP library is missing.

Each function restates one reference function and cites the file:line it follows
(paths are into the upstream reference tree, emanuele-artioli/elvis @ 2026-05-01).

Pinning status
--------------
* pinned by `tests/golden/*.npz` (outputs of the reference's own code, produced by
  `oracle/make_golden.py`): split/combine/stretch, chunk_for_devices,
  parallel_process_frames, resource_aware_restore, adaptive_restore,
  _extract_tile_with_halo, masked PSNR/MSE, strength-map npz codec.
* PARITY UNPINNED (cv2 is absent in the build image, no fixtures in the reference):
  `area_downscale_u8` (cv2.resize INTER_AREA), `nearest_upscale_map` (INTER_NEAREST),
  and therefore the cv2-dependent parts of `upscale_adaptive` (elvis.py:2522-2600),
  `blended_restoration` (utils.py:1575-1601).  Their control flow is restated from
  the source text; OpenCV's documented rounding rules are used.
"""
from __future__ import annotations

import math
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np


# --------------------------------------------------------------------------- blocks
def split_image_into_blocks(image: np.ndarray, block_size: int) -> np.ndarray:
    """elvis.py:1369-1385 - (H,W,C) -> (By,Bx,b,b,C) view; ValueError if not divisible."""
    h, w, c = image.shape
    if h % block_size != 0 or w % block_size != 0:
        raise ValueError("Image dimensions must be divisible by block_size.")
    by, bx = h // block_size, w // block_size
    return image.reshape(by, block_size, bx, block_size, c).swapaxes(1, 2)


def combine_blocks_into_image(blocks: np.ndarray) -> np.ndarray:
    """elvis.py:1429-1434 - inverse of split."""
    by, bx, b, _, c = blocks.shape
    return blocks.swapaxes(1, 2).reshape(by * b, bx * b, c)


def stretch_frame(shrunk: np.ndarray, binary_mask: np.ndarray, block_size: int) -> np.ndarray:
    """elvis.py:1436-1455 - place kept blocks where mask==0, zeros elsewhere."""
    by, bx = binary_mask.shape
    c = shrunk.shape[2]
    out = np.zeros((by, bx, block_size, block_size, c), dtype=shrunk.dtype)
    sb = split_image_into_blocks(shrunk, block_size)
    out[binary_mask == 0] = sb.reshape(-1, block_size, block_size, c)
    return combine_blocks_into_image(out)


# --------------------------------------------------------------------------- sharding
@dataclass
class ChunkSpec:
    """elvis.py:246-252."""
    start: int
    end: int
    device: object
    chunk_id: int = 0


def chunk_for_devices(total: int, devices: Sequence, min_chunk_size: int = 1) -> List[ChunkSpec]:
    """elvis.py:255-280 - contiguous split, first total%D devices get +1."""
    if not devices or total <= 0:
        return []
    nd = len(devices)
    base, rem = total // nd, total % nd
    chunks, start = [], 0
    for idx, dev in enumerate(devices):
        size = base + (1 if idx < rem else 0)
        if size < min_chunk_size and idx > 0:
            continue
        end = start + size
        if end > start:
            chunks.append(ChunkSpec(start, end, dev, idx))
        start = end
    return chunks


def parallel_process_frames(process_fn, frames, devices, chunk_size=None, max_workers=None):
    """elvis.py:283-353 - chunk, map, reassemble by ascending chunk_id."""
    if not frames:
        return []
    if not devices:
        devices = ["cpu"]
    n = len(frames)
    if chunk_size is None:
        chunks = chunk_for_devices(n, devices)
    else:
        chunks, cur, cid = [], 0, 0
        while cur < n:
            end = min(cur + chunk_size, n)
            chunks.append(ChunkSpec(cur, end, devices[cid % len(devices)], cid))
            cur, cid = end, cid + 1
    if not chunks:
        return []
    if len(chunks) == 1:
        c = chunks[0]
        return process_fn(frames[c.start:c.end], c.device)
    results: Dict[int, list] = {}
    workers = max_workers or min(len(chunks), len(devices))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        futs = [ex.submit(lambda c=c: (c.chunk_id, process_fn(frames[c.start:c.end], c.device))) for c in chunks]
        for f in futs:
            cid, out = f.result()
            results[cid] = out
    output = []
    for c in sorted(chunks, key=lambda c: c.chunk_id):
        output.extend(results[c.chunk_id])
    return output


# --------------------------------------------------------------------------- resize rules (UNPINNED)
def area_downscale_u8(img: np.ndarray, factor: int, rounding: str = "cv2") -> np.ndarray:
    """cv2.resize(img,(W/f,H/f),INTER_AREA) at an integer factor (elvis.py:2565, 2581).

    UNPINNED (no cv2 here).  Box mean over f x f.  rounding="cv2": OpenCV's u8 paths -
    the 2x2 fast path is (a+b+c+d+2)>>2, other integer factors are
    saturate_cast<uchar>(sum * (1.f/area)) i.e. float32 product rounded half-to-even.
    rounding="half_up": (sum + area/2) // area.
    """
    if factor == 1:
        return img.copy()
    h, w, c = img.shape
    if h % factor or w % factor:
        raise ValueError("area_downscale_u8 needs H,W divisible by the factor")
    s = img.reshape(h // factor, factor, w // factor, factor, c).astype(np.uint32).sum(axis=(1, 3))
    area = factor * factor
    if rounding == "half_up" or (rounding == "cv2" and factor == 2):
        return ((s + area // 2) // area).astype(np.uint8)
    if rounding == "cv2":
        prod = s.astype(np.float32) * np.float32(1.0 / area)
        return np.clip(np.rint(prod), 0, 255).astype(np.uint8)
    raise ValueError(rounding)


def nearest_upscale_map(dmap: np.ndarray, block_size: int) -> np.ndarray:
    """cv2.resize(map,(W,H),INTER_NEAREST) at the exact integer block factor == np.repeat
    (utils.py:1589).  UNPINNED."""
    return np.repeat(np.repeat(dmap, block_size, axis=0), block_size, axis=1)


# --------------------------------------------------------------------------- recompose variants
def recompose_select(a: np.ndarray, b: np.ndarray, pred: np.ndarray, block: int) -> np.ndarray:
    """out block (i,j) = pred[i,j] ? a : b  - the primitive behind elvis.py:2584-2595,
    elvis.py:2975-2978.  Pixels outside the floor(H/block) x floor(W/block) grid take b."""
    h, w, _ = a.shape
    by, bx = pred.shape
    m = np.zeros((h, w), dtype=bool)
    up = nearest_upscale_map(pred.astype(bool), block)
    m[: min(h, by * block), : min(w, bx * block)] = up[:h, :w]
    return np.where(m[:, :, None], a, b)


def upscale_adaptive(frame: np.ndarray, level_map: np.ndarray, block_size: int,
                     upsample_fn: Callable[[np.ndarray], np.ndarray], *, step: int = 2,
                     rounding: str = "cv2") -> np.ndarray:
    """elvis.py:2522-2600 (`upscale_realesrgan_adaptive`), staged coarse-to-fine recompose.

    step=2 follows the reference line by line: factors=2**map (2558); start at
    frame/max_factor (2565); per stage cur=upsample_fn(cur) (2575), ref=frame
    area-downscaled to cur's size (2581), block factor<=f ? ref : cur, else factor:=f
    (2584-2592 - note the in-place clamp of the map), f/=2 (2598).

    step=4 is the build's generalisation for a native 4x super-resolver (README.md:50):
    each stage multiplies resolution by 4 while f>=4... the stage factor sequence is
    max/4, max/16, ...; a trailing 2x stage (when log2(max) is odd) calls upsample_fn
    and area-halves its output.  With max_factor==4 it is a single 4x call followed by
    the f=1 paste `level==0 ? frame : SR`.
    """
    factors = np.power(2, level_map).astype(np.int32)
    max_factor = int(factors.max())
    h, w, _ = frame.shape
    cur = area_downscale_u8(frame, max_factor, rounding)
    by, bx = factors.shape
    f = max_factor
    while f > 1:
        if step == 2 or f < step:
            nf = f // 2
            up = upsample_fn(cur)
            if up.shape[0] != cur.shape[0] * 2:  # a 4x fn used for a 2x stage
                up = area_downscale_u8(up, up.shape[0] // (cur.shape[0] * 2), rounding)
        else:
            nf = f // step
            up = upsample_fn(cur)
        cur = up
        cb = block_size // nf
        ref = area_downscale_u8(frame, nf, rounding)
        pred = factors <= nf
        cur = recompose_select(ref, cur, pred, cb)
        factors = np.where(pred, factors, nf)
        f = nf
    return cur


def rounds_recompose(frames: List[np.ndarray], maps: np.ndarray, block_size: int,
                     restore_batch: Callable[[List[np.ndarray]], List[np.ndarray]],
                     batch_size: int = 4, max_rounds: Optional[int] = None) -> List[np.ndarray]:
    """elvis.py:2947-2981 - the iterative (InstantIR-slot) round loop.

    for r in range(max(map)): restore every frame that still has map>0; paste back the
    ORIGINAL decoded blocks where map<=0 (2975-2978); decrement positive entries (2980).
    `max_rounds` caps the loop (the build's single-pass Blur/DCT slot uses 1 with a map
    clamped to {0,1}).
    """
    cur = [f.copy() for f in frames]
    orig = [f.copy() for f in frames]
    m = np.asarray(maps, dtype=np.int32).copy()
    rounds = int(m.max()) if m.size else 0
    if max_rounds is not None:
        rounds = min(rounds, max_rounds)
    for _ in range(rounds):
        active = [i for i in range(len(cur)) if np.any(m[i] > 0)]
        if not active:
            break
        for off in range(0, len(active), max(1, batch_size)):
            idxs = active[off:off + max(1, batch_size)]
            outs = restore_batch([cur[i] for i in idxs])
            for i, o in zip(idxs, outs):
                done = m[i] <= 0
                cur[i] = recompose_select(orig[i], o, done, block_size)
        m[m > 0] -= 1
    return cur


def blend_by_map(orig: np.ndarray, rest: np.ndarray, dmap: np.ndarray, block_size: int, alpha: float) -> np.ndarray:
    """utils.py:1581-1599 - mask=(NEAREST-upsampled map>0); fp32 blend; clip; TRUNCATE to u8."""
    h, w = orig.shape[:2]
    by, bx = h // block_size, w // block_size
    if dmap.shape != (by, bx):   # utils.py:1586-1587: INTER_NEAREST resize of the map to the block grid
        ys0 = np.arange(by) * dmap.shape[0] // by
        xs0 = np.arange(bx) * dmap.shape[1] // bx
        dmap = dmap[ys0][:, xs0]
    # INTER_NEAREST to (w,h): dst x -> floor(x * bx / w)
    ys = (np.arange(h) * by // h).clip(0, by - 1)
    xs = (np.arange(w) * bx // w).clip(0, bx - 1)
    mask = (dmap.astype(np.float32)[ys][:, xs] > 0).astype(np.float32)[:, :, None]
    w_rest = mask * np.float32(alpha)
    w_orig = np.float32(1.0) - w_rest
    blended = orig.astype(np.float32) * w_orig + rest.astype(np.float32) * w_rest
    return np.clip(blended, 0, 255).astype(np.uint8)


def restore_video_adaptively(restore_fn, frames, degradation_maps, block_size=16, **kwargs):
    """presley.py:1219-1275 - one restore pass per distinct level, per-block pick;
    pixels outside the floored grid stay 0 (presley.py:1265)."""
    if not frames:
        return []
    h, w = frames[0].shape[:2]
    by, bx = h // block_size, w // block_size
    levels = set()
    for d in degradation_maps:
        levels.update(np.unique(d))
    versions = {}
    for level in sorted(levels):
        kw = dict(kwargs)
        kw["degradation_level"] = level
        res = restore_fn(frames=frames, **kw)
        versions[level] = res[0] if (isinstance(res, tuple) and len(res) == 2) else res
    out = []
    for i in range(len(frames)):
        ff = np.zeros((h, w, 3), dtype=np.uint8)
        d = degradation_maps[i]
        for y in range(by):
            for x in range(bx):
                sl = (slice(y * block_size, (y + 1) * block_size), slice(x * block_size, (x + 1) * block_size))
                ff[sl] = versions[d[y, x]][i][sl]
        out.append(ff)
    return out


# --------------------------------------------------------------------------- tiler
def resource_aware_restore(restore_fn, frames, tile_size=512, halo=16, chunk_size=8, chunk_overlap=2,
                           max_workers=1, device="cuda", **kwargs):
    """utils.py:176-326 - spatial tiles x temporal chunks, feathered fp32 accumulate,
    normalise, clip, TRUNCATING u8 cast.  Raises ValueError like the reference when an
    edge tile is thinner than halo//2 (numpy broadcast failure at utils.py:287-294)."""
    if not frames:
        return []
    h, w = frames[0].shape[:2]
    n = len(frames)
    do_tiling = tile_size > 0 and (h > tile_size or w > tile_size)
    do_chunking = chunk_size > 0 and n > chunk_size
    if not do_tiling and not do_chunking:
        return restore_fn(frames=frames, device=device, **kwargs)
    acc = [np.zeros((h, w, 3), np.float32) for _ in range(n)]
    wsum = [np.zeros((h, w, 1), np.float32) for _ in range(n)]
    if do_tiling:
        ys, xs = range(0, h, tile_size - halo), range(0, w, tile_size - halo)
    else:
        ys, xs, tile_size = [0], [0], max(h, w)
    if do_chunking:
        ts = range(0, n, chunk_size - chunk_overlap)
    else:
        ts, chunk_size = [0], n
    results = []
    for t0 in ts:
        for y0 in ys:
            for x0 in xs:
                t1, y1, x1 = min(t0 + chunk_size, n), min(y0 + tile_size, h), min(x0 + tile_size, w)
                chunk = [f[y0:y1, x0:x1] for f in frames[t0:t1]]
                try:
                    out = restore_fn(frames=chunk, device=device, tile_coords=(t0, t1, y0, y1, x0, x1), **kwargs)
                except Exception:  # utils.py:251-254 identity fallback
                    out = chunk
                results.append((t0, t1, y0, y1, x0, x1, out))
    for (t0, t1, y0, y1, x0, x1, out) in results:
        ch, cw = out[0].shape[:2]
        sw = np.ones((ch, cw, 1), np.float32)
        if do_tiling:
            fe = halo // 2
            if fe > 0:
                if y0 > 0:
                    sw[:fe, :, :] *= np.linspace(0, 1, fe)[:, None, None]
                if y1 < h:
                    sw[-fe:, :, :] *= np.linspace(1, 0, fe)[:, None, None]
                if x0 > 0:
                    sw[:, :fe, :] *= np.linspace(0, 1, fe)[None, :, None]
                if x1 < w:
                    sw[:, -fe:, :] *= np.linspace(1, 0, fe)[None, :, None]
        for i, fr in enumerate(out):
            gt = t0 + i
            tw = 1.0
            if do_chunking:
                if t0 > 0 and i < chunk_overlap:
                    tw *= (i + 1) / (chunk_overlap + 1)
                if t1 < n and i >= (len(out) - chunk_overlap):
                    tw *= (len(out) - i) / (chunk_overlap + 1)
            tot = sw * tw
            acc[gt][y0:y1, x0:x1] += fr.astype(np.float32) * tot
            wsum[gt][y0:y1, x0:x1] += tot
    final = []
    for i in range(n):
        safe = wsum[i].copy()
        safe[~(wsum[i] > 0)] = 1.0
        acc[i] /= safe
        final.append(np.clip(acc[i], 0, 255).astype(np.uint8))
    return final


def adaptive_restore(restore_fn, frames, degradation_maps=None, block_size=16, tile_coords=None,
                     threshold=0.0, **kwargs):
    """utils.py:329-394 - restore only where the tile's map slice exceeds `threshold`."""
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


def extract_tile_with_halo(frame, y, x, tile_h, tile_w, halo):
    """utils.py:1227-1250."""
    h, w = frame.shape[:2]
    y0, x0 = max(0, y - halo), max(0, x - halo)
    y1, x1 = min(h, y + tile_h + halo), min(w, x + tile_w + halo)
    tile = frame[y0:y1, x0:x1].copy()
    ct, cl = y - y0, x - x0
    return tile, (ct, cl, ct + tile_h, cl + tile_w)


# --------------------------------------------------------------------------- metrics
def masked_mse(ref, dec, mask=None) -> float:
    """elvis.py:653-671."""
    r, d = ref.astype(np.float32), dec.astype(np.float32)
    if mask is not None:
        v = mask.astype(bool)
        if not np.any(v):
            return 0.0
        diff = r[v] - d[v]
    else:
        diff = r - d
    return float(np.mean(diff ** 2)) if diff.size else 0.0


def masked_psnr(ref, dec, mask=None) -> float:
    """elvis.py:627-650 - 20 log10(255/sqrt(mse)), capped at 100 dB."""
    if mask is not None and not np.any(mask.astype(bool)):
        return 100.0
    mse = masked_mse(ref, dec, mask)
    if mse < 1e-10:
        return 100.0
    return float(min(20 * math.log10(255.0 / math.sqrt(mse)), 100.0))


def psnr_whole(ref, dec, data_range=255.0) -> float:
    """presley.py:235-245 - 10 log10(range^2/mse), inf at 0 (the parity-report PSNR)."""
    mse = np.mean((ref.astype(np.float32) - dec.astype(np.float32)) ** 2)
    return float("inf") if mse == 0 else float(10 * np.log10((data_range ** 2) / mse))


def block_ssim(f1: np.ndarray, f2: np.ndarray, block_size: int) -> np.ndarray:
    """utils.py:572-608 - per-block `pytorch_msssim.ssim(..., data_range=1.0, size_average=False)`.

    PARITY UNPINNED: pytorch_msssim is absent; this restates its published algorithm in float64:
    11-tap Gaussian window (sigma 1.5, normalised), separable "valid" smoothing that is SKIPPED along a dimension
    shorter than the window, C1 = 0.01^2, C2 = 0.03^2, ssim_map = luminance * contrast-structure, mean over the
    map, then over channels."""
    h, w = f1.shape[:2]
    by, bx = h // block_size, w // block_size
    coords = np.arange(11, dtype=np.float32) - 5
    g = np.exp(-(coords ** 2) / np.float32(2 * 1.5 ** 2)).astype(np.float32)
    g = (g / g.sum()).astype(np.float64)
    C1, C2 = 0.01 ** 2, 0.03 ** 2

    def smooth(x):   # x: (b, b)
        if x.shape[0] >= 11:
            x = np.stack([(x[i:i + 11] * g[:, None]).sum(0) for i in range(x.shape[0] - 10)])
        if x.shape[1] >= 11:
            x = np.stack([(x[:, j:j + 11] * g[None, :]).sum(1) for j in range(x.shape[1] - 10)], axis=1)
        return x

    out = np.zeros((by, bx), np.float64)
    a = f1.astype(np.float32).astype(np.float64) / 255.0
    b = f2.astype(np.float32).astype(np.float64) / 255.0
    for iy in range(by):
        for ix in range(bx):
            vals = []
            for c in range(f1.shape[2]):
                x = a[iy * block_size:(iy + 1) * block_size, ix * block_size:(ix + 1) * block_size, c]
                y = b[iy * block_size:(iy + 1) * block_size, ix * block_size:(ix + 1) * block_size, c]
                mu1, mu2 = smooth(x), smooth(y)
                s1, s2, s12 = smooth(x * x) - mu1 * mu1, smooth(y * y) - mu2 * mu2, smooth(x * y) - mu1 * mu2
                cs = (2 * s12 + C2) / (s1 + s2 + C2)
                vals.append((((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs).mean())
            out[iy, ix] = np.mean(vals)
    return out.astype(np.float32)
