"""Frames-in / frames-out restoration drivers: the reference's P1/P2/P3 protocols backed by the
MI355X SinSR path.

  P1  `upsample_fn(img_bgr_u8) -> img_bgr_u8`  (elvis.py:2528, 2575)  -> `get_sinsr_upsample_fn`
  P2  `restore_frames_sinsr(frames, downscale_maps, block_size, device, **kw)`
      = drop-in for `restore_frames_realesrgan` (elvis.py:2640-2682)
  P3  `restore_with_sinsr_naive(frames=, device=, **kw)` resolution-preserving restore_fn
      (signature template utils.py:1428-1473)

Model handles are process-global, cached per (device, params) under a lock like
`get_realesrgan_upsampler` (elvis.py:2607-2637).
"""
from __future__ import annotations

import threading
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from . import ops
from .recompose import (frames_to_device, frames_to_host, maps_to_device, rounds_recompose_device,
                        upscale_adaptive_device)
from .sinsr import SinSRModel
from .weights import SinSRConfig

_MODEL_CACHE: Dict[tuple, tuple] = {}
_MODEL_LOCK = threading.Lock()
DEFAULT_SEED = 42  # the reference's default sampler seed (elvis.py:89, utils.py:103)


def get_sinsr_model(device, *, cfg: Optional[SinSRConfig] = None, fp32: bool = False, weight_seed: int = 0,
                    state_dict=None, fuse_gn: bool = True, precision: Optional[str] = None) -> SinSRModel:
    """Get or create the cached SinSR runtime for `device` (thread-safe, never freed - the
    reference's cache policy, elvis.py:2611-2637)."""
    dev = torch.device(device)
    L.require_gpu(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    cfg = cfg or SinSRConfig()
    # the key holds `cfg` itself (a frozen dataclass) and the identity of the caller's state_dict; the
    # cache entry keeps a strong reference to that dict, so its id cannot be reused by another object
    key = (str(dev), cfg, bool(fp32), precision, int(weight_seed), id(state_dict) if state_dict is not None else 0, bool(fuse_gn))
    with _MODEL_LOCK:
        hit = _MODEL_CACHE.get(key)
        if hit is None:
            try:
                m = SinSRModel(cfg, state_dict, dev, torch.float32 if fp32 else torch.float16, weight_seed, fuse_gn,
                               precision=precision)
            except RuntimeError as exc:
                raise RuntimeError(f"SinSR failed on {dev}: {exc}") from exc
            hit = _MODEL_CACHE[key] = (m, state_dict)
    return hit[0]


def sr4x_device(model: SinSRModel, lr_d: torch.Tensor, frame_indices: Sequence[int], seed: int = DEFAULT_SEED,
                swap_rb: bool = True, batch: int = 1) -> torch.Tensor:
    """[n,h,w,3] u8 -> [n,4h,4w,3] u8 on device; sampler noise keyed on the global frame index."""
    n, h, w, _ = lr_d.shape
    outs = []
    with torch.cuda.device(model.device):
        for s in range(0, n, batch):
            idx = list(frame_indices[s:s + batch])
            noise = model.make_noise(seed, idx, h, w)
            outs.append(model.forward(lr_d[s:s + batch].contiguous(), noise, swap_rb=swap_rb))
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


def get_sinsr_upsample_fn(device, *, scale: int = 2, seed: int = DEFAULT_SEED, frame_index: int = 0,
                          fp32: bool = False, cfg: Optional[SinSRConfig] = None) -> Callable[[np.ndarray], np.ndarray]:
    """P1: "a single 2x upscale" of a BGR uint8 HWC image (elvis.py:2528, 2548, 2575).  scale=2
    runs the 4x network and area-halves the result (what RealESRGANer.enhance(outscale=2) does
    with its x4 net, elvis.py:2515); scale=4 returns the native 4x output."""
    if scale not in (2, 4):
        raise ValueError("scale must be 2 or 4")
    model = get_sinsr_model(device, cfg=cfg, fp32=fp32)

    def upsample_fn(img: np.ndarray) -> np.ndarray:
        try:
            with torch.cuda.device(model.device):   # pool threads start on device 0 (P2, elvis.py:342-346)
                d = frames_to_device([img], model.device)
                out = sr4x_device(model, d, [frame_index], seed)
                if scale == 2:
                    out = ops.area_downscale_u8(out, 2)
                return frames_to_host(out)[0]
        except RuntimeError as exc:
            raise RuntimeError(f"SinSR failed on {model.device}: {exc}") from exc

    return upsample_fn


def restore_clip_single4x_device(model: SinSRModel, frames_d: torch.Tensor, levels_d: torch.Tensor, block_size: int,
                                 frame_indices: Sequence[int], seed: int = DEFAULT_SEED, swap_rb: bool = True,
                                 noise: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                                 batch: int = 6, active: Optional[Sequence[bool]] = None) -> torch.Tensor:
    """The north-star Downsample path, fully on device: whole frame /4 (INTER_AREA, elvis.py:2565)
    -> ONE SinSR 4x call (README.md:50) -> final-stage paste of elvis.py:2584-2595 at f=1
    (`level == 0 ? decoded frame : SR`).  Frames whose map is all zero skip the network.
    `noise` ([n,3,Hp,Wp] f32, resident) may be passed to keep host RNG out of a timed region.
    `batch` frames go through the network together (fills the GPU better on the UNet's small
    levels; results are identical - every op is per-sample).  `active[i]` (does frame i have any
    level > 0) may be passed from the host copy of the map to avoid a device->host sync here."""
    n, H, W, _ = frames_d.shape
    if H % 4 or W % 4 or H % block_size or W % block_size:
        raise ValueError("Image dimensions must be divisible by block_size and by 4.")
    if out is None:
        out = torch.empty_like(frames_d)
    if active is None:
        active = (levels_d > 0).flatten(1).any(dim=1).tolist()
    # frames per network invocation, capped by pixels: 15 frames of 1080p is the largest footprint measured
    # (~8 GB per 128-channel f16 tensor, a few of them live); results do not depend on it
    batch = max(1, min(int(batch), (15 * 1080 * 1920 * (2 if model.dtype == torch.float16 else 1)) // (2 * H * W)))
    with torch.cuda.device(model.device):
        todo = [i for i in range(n) if active[i]]
        for i in range(n):
            if not active[i]:
                out[i:i + 1] = frames_d[i:i + 1]
        for s0 in range(0, len(todo), batch):
            idx = todo[s0:s0 + batch]
            contiguous = idx == list(range(idx[0], idx[0] + len(idx)))
            sel = slice(idx[0], idx[0] + len(idx)) if contiguous else torch.tensor(idx, device=frames_d.device)
            f = frames_d[sel] if contiguous else frames_d[sel].contiguous()
            lv = levels_d[sel] if contiguous else levels_d[sel].contiguous()
            lr = ops.area_downscale_u8(f, 4)
            if noise is not None:
                nz = noise[sel] if contiguous else noise[sel].contiguous()
            else:
                nz = model.make_noise(seed, [frame_indices[i] for i in idx], H // 4, W // 4)
            sr = model.forward(lr, nz, swap_rb=swap_rb)
            if contiguous:
                ops.recompose_u8(f, sr, lv, block_size, 0, out=out[sel])
            else:
                out[sel] = ops.recompose_u8(f, sr, lv, block_size, 0)
    return out


class _HostClipPipeline:
    """Persistent staging for `restore_clip_single4x_host`: device buffers for one clip, a pinned host
    buffer for the sampler noise, two copy streams and a small thread pool for the noise generator."""

    def __init__(self, model: SinSRModel, n: int, H: int, W: int, by: int, bx: int):
        from concurrent.futures import ThreadPoolExecutor
        dev = model.device
        hp, wp = model.padded_latent_shape(H // 4, W // 4)
        self.shape = (n, H, W, by, bx)
        self.frames_d = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
        self.out_d = torch.empty_like(self.frames_d)
        self.levels_d = torch.empty((n, by, bx), dtype=torch.int32, device=dev)
        self.noise_d = torch.empty((n, model.cfg.latent_ch, hp, wp), dtype=torch.float32, device=dev)
        # two pinned noise buffers, used by alternate calls: a call returns after ENQUEUEING, so its noise uploads may
        # still be pending when the next call's generator threads start writing - they write the other buffer, and
        # `noise_up[k]` (recorded after a call's last upload from buffer k) is waited for before buffer k is rewritten
        self.noise_hs = [torch.empty((n, model.cfg.latent_ch, hp, wp), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.noise_up = [None, None]
        self.calls = 0
        self.h2d, self.d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        self.pool = ThreadPoolExecutor(max_workers=8, thread_name_prefix="elvis-noise")


def restore_clip_single4x_host(model: SinSRModel, frames_h: torch.Tensor, levels_h: torch.Tensor, block_size: int,
                               frame_indices: Sequence[int], out_h: Optional[torch.Tensor] = None,
                               seed: int = DEFAULT_SEED, swap_rb: bool = True, batch: int = 15,
                               want_device: bool = False):
    """Host-to-host form of `restore_clip_single4x_device` (the metric's timed region):
    `frames_h` [n,H,W,3] uint8 and `levels_h` [n,By,Bx] int32 in host memory (pinned for full-rate
    PCIe) -> restored frames in `out_h` (host, pinned).  Per batch of `batch` frames: upload on a copy
    stream, sampler noise generated by a thread pool (torch CPU generator keyed on the global frame
    index, exactly `weights.frame_noise`) into pinned memory and uploaded, network + recompose on the
    compute stream, download on a second copy stream - so transfers and the noise generator overlap the
    previous / next batch's compute.  Returns after everything has been ENQUEUED; synchronise the device
    (or `out_h`'s consumers) before reading `out_h`.  `want_device=True` also returns the restored clip's
    HBM buffer (valid until the next call with the same model and shape) for a following all-gather."""
    n, H, W, _ = frames_h.shape
    by, bx = levels_h.shape[1:]
    if H % 4 or W % 4 or H % block_size or W % block_size:
        raise ValueError("Image dimensions must be divisible by block_size and by 4.")
    if frames_h.is_cuda or levels_h.is_cuda or frames_h.dtype != torch.uint8 or levels_h.dtype != torch.int32:
        raise ValueError("restore_clip_single4x_host takes host uint8 frames and host int32 maps")
    if out_h is None:
        out_h = torch.empty((n, H, W, 3), dtype=torch.uint8).pin_memory()
    key = (n, H, W, by, bx)
    with _MODEL_LOCK:
        # the staging buffers live ON the model object (freed with it), keyed by clip shape
        pipes = model.__dict__.setdefault("_host_pipelines", {})
        pl = pipes.get(key)
        if pl is None:
            with torch.cuda.device(model.device):
                pl = pipes[key] = _HostClipPipeline(model, n, H, W, by, bx)
    active = (levels_h > 0).flatten(1).any(dim=1).tolist()
    with _MODEL_LOCK:
        slot = pl.calls & 1
        pl.calls += 1
    noise_h = pl.noise_hs[slot]
    if pl.noise_up[slot] is not None:
        pl.noise_up[slot].synchronize()   # the call before last uploaded from this buffer: done before it is rewritten
    cfg, (hp, wp) = model.cfg, noise_h.shape[2:]

    def gen(i):   # bit-identical to weights.frame_noise(cfg, seed, frame_indices[i], hp, wp)
        g = torch.Generator().manual_seed(int(seed) * 1000003 + int(frame_indices[i]))
        torch.randn((1, cfg.latent_ch, hp, wp), generator=g, dtype=torch.float32, out=noise_h[i:i + 1])

    step = max(1, batch)
    futures = {i: pl.pool.submit(gen, i) for i in range(n) if active[i]}
    with torch.cuda.device(model.device):
        compute = torch.cuda.current_stream(model.device)
        pl.h2d.wait_stream(compute)   # the previous call's kernels may still read the staging buffers
        pl.d2h.wait_stream(compute)
        for s0 in range(0, n, step):
            sel = slice(s0, min(s0 + step, n))
            for i in range(sel.start, sel.stop):
                if i in futures:
                    futures[i].result()
            with torch.cuda.stream(pl.h2d):
                pl.frames_d[sel].copy_(frames_h[sel], non_blocking=True)
                pl.levels_d[sel].copy_(levels_h[sel], non_blocking=True)
                pl.noise_d[sel].copy_(noise_h[sel], non_blocking=True)
                if sel.stop >= n:
                    pl.noise_up[slot] = torch.cuda.Event()
                    pl.noise_up[slot].record(pl.h2d)
            compute.wait_stream(pl.h2d)
            restore_clip_single4x_device(model, pl.frames_d[sel], pl.levels_d[sel], block_size,
                                         list(frame_indices[sel]), seed, swap_rb, noise=pl.noise_d[sel],
                                         out=pl.out_d[sel], batch=step, active=active[sel])
            pl.d2h.wait_stream(compute)
            with torch.cuda.stream(pl.d2h):
                out_h[sel].copy_(pl.out_d[sel], non_blocking=True)
        compute.wait_stream(pl.d2h)   # a device synchronize / later kernel now also covers the downloads
    return (out_h, pl.out_d) if want_device else out_h


def restore_frames_sinsr(frames: List[np.ndarray], downscale_maps: np.ndarray, block_size: int, device,
                         *, seed: int = DEFAULT_SEED, first_frame_index: int = 0, fp32: bool = False,
                         schedule: str = "staged", staged_2x: bool = False, cfg: Optional[SinSRConfig] = None,
                         precision: Optional[str] = None, **_ignored) -> List[np.ndarray]:
    """Pure restoration function (no file IO, no parallelisation), drop-in for
    `restore_frames_realesrgan` (elvis.py:2640-2682): BGR uint8 frames + per-block log2
    downscale maps -> restored frames.

    schedule="staged" (default: the reference's semantics): the coarse-to-fine loop of
    elvis.py:2570-2598 generalised to 4x stages (`staged_2x=True`: the reference's 2x-per-stage loop
    with the 4x net area-halved per stage).  schedule="single4x" (the north-star / benchmark path): one
    SinSR 4x call from the /4 level per frame, whatever the map's maximum (README.md:50) - blocks of
    level 1 lose the detail the staged loop would keep at the /2 stage.
    Unknown model kwargs of the reference call (model_name, tile, ...) are accepted and ignored
    (the `**kwargs` convention of the P3 surface, utils.py:1428).
    """
    if not frames:
        return []
    dev = torch.device(device)
    model = get_sinsr_model(dev, cfg=cfg, fp32=fp32, precision=precision)
    with torch.cuda.device(model.device):
        frames_d = frames_to_device(frames, model.device)
        n = frames_d.shape[0]
        maps_d = maps_to_device(downscale_maps, n, model.device)
        if schedule == "single4x":
            gidx = [first_frame_index + i for i in range(n)]
            return frames_to_host(restore_clip_single4x_device(model, frames_d, maps_d, block_size, gidx, seed))
        if schedule != "staged":
            raise ValueError(f"unknown schedule '{schedule}'")
        out = torch.empty_like(frames_d)
        # the reference derives the stage schedule from each frame's own max level
        # (elvis.py:2561): group frames by it so a batch shares one schedule.
        maxlv = maps_d.flatten(1).max(dim=1).values.tolist()
        for lv in sorted(set(maxlv)):
            idx = [i for i, v in enumerate(maxlv) if v == lv]
            if lv == 0:
                out[idx] = frames_d[idx]
                continue
            it = torch.tensor(idx, device=model.device)
            sub_f, sub_m = frames_d[it].contiguous(), maps_d[it].contiguous()
            gidx = [first_frame_index + i for i in idx]

            def up(cur, _g=gidx):
                return sr4x_device(model, cur, _g, seed)

            out[it] = upscale_adaptive_device(sub_f, sub_m, block_size, up, sr_scale=2 if staged_2x else 4)
        return frames_to_host(out)


def restore_with_sinsr_naive(frames: List[np.ndarray], device="cuda", seed: int = DEFAULT_SEED,
                             tile_coords=None, degradation_level=None, **kwargs) -> List[np.ndarray]:
    """P3 restore_fn: resolution-preserving whole-frame restore of RGB uint8 frames - /4 area
    downscale, SinSR 4x back to size (the shape of restore_with_realesrgan_naive, utils.py:1428,
    which runs the 4x net and resizes back).  Accepts and ignores unknown kwargs."""
    if not frames:
        return []
    dev = torch.device("cuda:0" if str(device) == "cuda" else device)
    model = get_sinsr_model(dev, fp32=bool(kwargs.get("fp32", False)), cfg=kwargs.get("cfg"))
    t0 = tile_coords[0] if tile_coords else 0
    with torch.cuda.device(model.device):
        d = frames_to_device(frames, model.device)
        n, h, w, _ = d.shape
        ph, pw = (-h) % 4, (-w) % 4
        if ph or pw:
            raise ValueError("restore_with_sinsr_naive needs H,W divisible by 4")
        lr = ops.area_downscale_u8(d, 4)
        out = sr4x_device(model, lr, [t0 + i for i in range(n)], seed, swap_rb=False)
        return frames_to_host(out)


def restore_frames_rounds(frames: List[np.ndarray], maps: np.ndarray, block_size: int, device,
                          restore_dev: Callable[[torch.Tensor], torch.Tensor], batch_size: int = 4,
                          max_rounds: Optional[int] = None) -> List[np.ndarray]:
    """Frames-level form of the iterative round loop (elvis.py:2947-2981) for any on-device
    restorer (`restore_dev`: [n,H,W,3] u8 -> same) - the slot the Blur / DCT restorers fill."""
    if not frames:
        return []
    dev = torch.device(device)
    L.require_gpu(dev)
    with torch.cuda.device(dev):
        frames_d = frames_to_device(frames, dev)
        maps_d = maps_to_device(maps, frames_d.shape[0], dev)
        return frames_to_host(rounds_recompose_device(frames_d, maps_d, block_size, restore_dev, batch_size, max_rounds))


# ----------------------------------------------------------------------------- Blur / DCT slots
_RESTORER_CACHE: Dict[tuple, tuple] = {}


def _get_restorer(kind: str, device, fp32: bool, cfg=None, state_dict=None):
    from .restorers import DCNRestorer, SwinDeblur
    dev = torch.device(device)
    L.require_gpu(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (kind, str(dev), bool(fp32), cfg, id(state_dict) if state_dict is not None else 0)
    with _MODEL_LOCK:
        hit = _RESTORER_CACHE.get(key)
        m = hit[0] if hit is not None else None
        if m is None:
            cls = SwinDeblur if kind == "blur" else DCNRestorer
            args = (cfg,) if cfg is not None else ()
            try:
                m = cls(*args, state_dict=state_dict, device=dev, dtype=torch.float32 if fp32 else torch.float16)
            except RuntimeError as exc:
                raise RuntimeError(f"{cls.__name__} failed on {dev}: {exc}") from exc
            _RESTORER_CACHE[key] = (m, state_dict)   # the strong reference keeps id(state_dict) unique
    return m


def restore_frames_blur(frames: List[np.ndarray], blur_maps: np.ndarray, block_size: int, device, *,
                        batch_size: int = 4, max_rounds: Optional[int] = None, fp32: bool = False, cfg=None,
                        state_dict=None, **_ignored) -> List[np.ndarray]:
    """ELVIS v2 Blur client side (the loop of `_instantir_chunk_worker`, elvis.py:2947-2981) with the
    SwinTormer-style deblurrer in the model slot: for r in range(max(map)) restore every frame that
    still has map > 0, re-paste the ORIGINAL decoded blocks whose remaining level is <= 0,
    decrement.  BGR uint8 frames in and out; `batch_size` as in the reference (elvis.py:91)."""
    if not frames:
        return []
    model = _get_restorer("blur", device, fp32, cfg, state_dict)
    return restore_frames_rounds(frames, blur_maps, block_size, model.device,
                                 lambda d: model.restore(d, swap_rb=True), batch_size, max_rounds)


def restore_frames_dct(frames: List[np.ndarray], strength_maps: np.ndarray, block_size: int, device, *,
                       fp32: bool = False, cfg=None, state_dict=None, **_ignored) -> List[np.ndarray]:
    """ELVIS v2 DCT client side.  The reference has no code for this slot; the build
    defines it with the round loop's signature and ONE pass: the LaplacianVCAR-style DCN restorer
    sees the whole chunk (its temporal window crosses frames), then `level > 0 ? restored : decoded`."""
    if not frames:
        return []
    model = _get_restorer("dct", device, fp32, cfg, state_dict)
    dev = model.device
    with torch.cuda.device(dev):
        frames_d = frames_to_device(frames, dev)
        maps_d = maps_to_device(strength_maps, frames_d.shape[0], dev)
        restored = model.restore(frames_d)
        return frames_to_host(ops.recompose_u8(frames_d, restored, maps_d, block_size, 0))


# ----------------------------------------------------------------------------- Blur / DCT slots, host to host
class _HostSlotPipeline:
    """Persistent staging for `restore_clip_slot_host`: the decoded clip, the working / output clip and the maps in
    HBM, one upload and one download stream."""

    def __init__(self, dev, n: int, H: int, W: int, by: int, bx: int):
        self.frames_d = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
        self.out_d = torch.empty_like(self.frames_d)
        self.maps_d = torch.empty((n, by, bx), dtype=torch.int32, device=dev)
        self.h2d, self.d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def restore_clip_slot_host(kind: str, model, frames_h: torch.Tensor, maps_h: torch.Tensor, block_size: int,
                           out_h: Optional[torch.Tensor] = None, *, batch_size: int = 2, upload_chunk: int = 6,
                           max_rounds: Optional[int] = None, swap_rb: bool = True, want_device: bool = False):
    """Host-to-host form of the Blur (`kind="blur"`: the round loop of elvis.py:2947-2981 around the Swin deblurrer) and
    DCT (`kind="dct"`: one pass of the DCN restorer, `level > 0 ? restored : decoded`) client paths - the timed region
    SURVEY.md 8(d) asks for on BASELINE configs 4 and 3.  `frames_h` [n,H,W,3] uint8 and `maps_h` [n,By,Bx] int32 in
    (pinned) host memory -> restored frames in `out_h` (host, pinned).  The clip is uploaded `upload_chunk` frames at a
    time on a copy stream, each chunk is restored as soon as it (and, for the DCT restorer's temporal window, the
    chunk after it) has arrived, and goes back on a second copy stream while the next chunk computes.

    Same frames as `restore_frames_blur` / `restore_frames_dct` give: the round loop runs chunk by chunk instead of
    round by round (frames are independent, so the order is free), and "re-paste where the remaining level <= 0 after
    r decrements" is evaluated as `map <= r` on the original map - no map is rewritten and nothing is read back from
    the device inside the call.  Returns after everything has been ENQUEUED; synchronise before reading `out_h`."""
    if kind not in ("blur", "dct"):
        raise ValueError(f"unknown slot '{kind}'")
    n, H, W, _ = frames_h.shape
    by, bx = maps_h.shape[1:]
    if H % block_size or W % block_size:
        raise ValueError("Image dimensions must be divisible by block_size.")
    if frames_h.is_cuda or maps_h.is_cuda or frames_h.dtype != torch.uint8 or maps_h.dtype != torch.int32 or maps_h.shape[0] != n:
        raise ValueError("restore_clip_slot_host takes host uint8 frames [n,H,W,3] and host int32 maps [n,By,Bx]")
    if out_h is None:
        out_h = torch.empty((n, H, W, 3), dtype=torch.uint8).pin_memory()
    dev = model.device
    key = (kind, n, H, W, by, bx)
    with _MODEL_LOCK:
        pipes = model.__dict__.setdefault("_host_pipelines", {})
        pl = pipes.get(key)
        if pl is None:
            with torch.cuda.device(dev):
                pl = pipes[key] = _HostSlotPipeline(dev, n, H, W, by, bx)
    peak = maps_h.flatten(1).max(dim=1).values.tolist() if n else []      # rounds each frame needs (host copy of the map)
    step = max(1, int(upload_chunk))
    halo = model.cfg.radius if kind == "dct" else 0
    bounds = [(s, min(s + step, n)) for s in range(0, n, step)]
    with torch.cuda.device(dev):
        compute = torch.cuda.current_stream(dev)
        pl.h2d.wait_stream(compute)            # the previous call's kernels may still read the staging buffers
        pl.d2h.wait_stream(compute)
        arrived = []
        with torch.cuda.stream(pl.h2d):
            for s, e in bounds:
                pl.frames_d[s:e].copy_(frames_h[s:e], non_blocking=True)
                pl.maps_d[s:e].copy_(maps_h[s:e], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(pl.h2d)
                arrived.append(ev)
        for k, (s, e) in enumerate(bounds):
            need = k
            while need + 1 < len(bounds) and bounds[need][1] < min(n, e + halo):
                need += 1                      # the temporal window of this chunk's last frames
            compute.wait_event(arrived[need])
            if kind == "dct":
                restored = model.restore(pl.frames_d, chunk=batch_size, frame_range=(s, e))
                ops.recompose_u8(pl.frames_d[s:e], restored, pl.maps_d[s:e], block_size, 0, out=pl.out_d[s:e])
            else:
                pl.out_d[s:e].copy_(pl.frames_d[s:e])
                rounds = max([int(v) for v in peak[s:e]] + [0])
                if max_rounds is not None:
                    rounds = min(rounds, max_rounds)
                for r in range(rounds):
                    act = [i for i in range(s, e) if peak[i] > r]
                    for off in range(0, len(act), max(1, batch_size)):
                        idx = act[off:off + max(1, batch_size)]
                        if idx == list(range(idx[0], idx[0] + len(idx))):
                            sel = slice(idx[0], idx[0] + len(idx))
                            restored = model.restore(pl.out_d[sel], swap_rb=swap_rb)
                            ops.recompose_u8(pl.frames_d[sel], restored, pl.maps_d[sel], block_size, r, out=pl.out_d[sel])
                        else:
                            it = torch.tensor(idx, device=dev)
                            restored = model.restore(pl.out_d[it].contiguous(), swap_rb=swap_rb)
                            pl.out_d[it] = ops.recompose_u8(pl.frames_d[it].contiguous(), restored, pl.maps_d[it].contiguous(),
                                                            block_size, r)
            pl.d2h.wait_stream(compute)
            with torch.cuda.stream(pl.d2h):
                out_h[s:e].copy_(pl.out_d[s:e], non_blocking=True)
        compute.wait_stream(pl.d2h)
    return (out_h, pl.out_d) if want_device else out_h
