"""Directory-level restoration drivers: the two functions `run_elvis` calls on the client side
(elvis.py:4722, 4794) with the MI355X restorers in the model slots.

  restore_downsampled_with_sinsr   <- restore_downsampled_with_realesrgan  (elvis.py:2685-2769)
  restore_blur_adaptive            <- restore_with_instantir_adaptive      (elvis.py:3000-3160)
  restore_dct_adaptive             -  the DCT slot the reference never wrote, given the
                                      Blur driver's shape

Same arguments, file naming and errors as the reference: a directory of PNG frames (BGR on read) and a
`(frames, blocks_y, blocks_x)` map; the Downsample driver writes `output_dir/<same names>`, the Blur /
DCT drivers rewrite the frames in place.  Frames are dealt to the devices by the `chunk_for_devices`
rule.  One device: in-process.  Several devices: one spawned process per GPU (what
`restore_with_instantir_adaptive` does, elvis.py:3124-3158), each reading its own frame range and
writing its own files - the file system is the gather, exactly as in the reference (elvis.py:2983-2985).
Sampler noise is keyed on the global frame index and the DCT restorer reads its temporal halo frames
from the directory, so results do not depend on the device count.
"""
from __future__ import annotations

import multiprocessing
import os
from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import torch

from .frameio import clear_directory, get_frame_paths, load_frame, save_frame
from .sharding import ChunkSpec, chunk_for_devices, resolve_device_list

DeviceSpec = Union[int, str, torch.device]
# (frames, maps, block_size, device, first_frame_index, **kw) -> frames ; replaceable for host-only tests
ShardFn = Callable[..., List[np.ndarray]]


def _device_str(dev: torch.device) -> str:
    return f"cuda:{dev.index or 0}" if dev.type == "cuda" else str(dev)


def _sinsr_shard(frames, maps, block_size, device, first_frame_index, **kw):
    from .restore import restore_frames_sinsr
    return restore_frames_sinsr(frames, maps, block_size, device, first_frame_index=first_frame_index, **kw)


def _blur_shard(frames, maps, block_size, device, first_frame_index, **kw):
    from .restore import restore_frames_blur
    return restore_frames_blur(frames, maps, block_size, device, **kw)


def _dct_shard(frames, maps, block_size, device, first_frame_index, **kw):
    from .restore import restore_frames_dct
    return restore_frames_dct(frames, maps, block_size, device, **kw)


def _shard_worker(shard_fn: ShardFn, in_dir: str, out_dir: str, names: Sequence[str], start: int, end: int,
                  maps: np.ndarray, block_size: int, device_str: str, halo: int, kw: dict) -> None:
    """Restore frames [start, end) of the sorted file list `names` on one device and write them to
    `out_dir` under the same names.  `halo` extra frames on each side are read (never written) for
    restorers with a temporal window; their maps are taken as given."""
    device = torch.device(device_str)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    lo, hi = max(0, start - halo), min(len(names), end + halo)
    frames = [load_frame(os.path.join(in_dir, names[i])) for i in range(lo, hi)]
    restored = shard_fn(frames, np.asarray(maps[lo:hi]), block_size, device, lo, **kw)
    if len(restored) != hi - lo:
        raise RuntimeError(f"restorer returned {len(restored)} frames for {hi - lo} inputs on {device_str}")
    for i in range(start, end):
        save_frame(restored[i - lo], os.path.join(out_dir, names[i]))


def _shard_process(err_path: str, job: tuple) -> None:
    """Entry point of a spawned per-GPU worker: a failure is written to stderr and to `err_path` as
    `device / [start, end) / traceback`, so the parent can say WHICH shard failed and why."""
    import sys
    import traceback
    try:
        _shard_worker(*job)
    except BaseException:
        start, end, device_str = job[4], job[5], job[8]
        msg = f"shard on {device_str}, frames [{start}, {end}):\n{traceback.format_exc()}"
        sys.stderr.write(msg)
        try:
            with open(err_path, "w") as f:
                f.write(msg)
        finally:
            raise SystemExit(1)


def _run_shards(shard_fn: ShardFn, in_dir: str, out_dir: str, names: List[str], chunks: List[ChunkSpec],
                maps: np.ndarray, block_size: int, halo: int, kw: dict) -> None:
    jobs = [(shard_fn, in_dir, out_dir, names, c.start, c.end, maps, block_size, _device_str(c.device), halo, kw)
            for c in chunks]
    if len(jobs) == 1:
        _shard_worker(*jobs[0])
        return
    import tempfile
    ctx = multiprocessing.get_context("spawn")   # never fork a process that may hold a GPU context
    with tempfile.TemporaryDirectory(prefix="elvis_shards_") as errdir:
        errs = [os.path.join(errdir, f"shard{i}.err") for i in range(len(jobs))]
        procs = [ctx.Process(target=_shard_process, args=(e, job)) for e, job in zip(errs, jobs)]
        for p in procs:
            p.start()
        failed = []
        for p, e, job in zip(procs, errs, jobs):
            p.join()
            if p.exitcode not in (0, None):
                detail = open(e).read() if os.path.exists(e) else \
                    f"shard on {job[8]}, frames [{job[4]}, {job[5]}): exit code {p.exitcode} (no traceback: killed?)"
                failed.append(detail)
    if failed:
        raise RuntimeError("restoration worker(s) failed:\n" + "\n".join(failed))


def _frames_and_maps(frames_dir: str, maps, what: str):
    paths = get_frame_paths(frames_dir)
    if not paths:
        raise ValueError(f"No frames found in {frames_dir}")
    maps = np.asarray(maps)
    if maps.ndim != 3 or maps.shape[0] != len(paths):
        raise ValueError(f"{what} length ({maps.shape[0] if maps.ndim else 0}) does not match frame count ({len(paths)}).")
    return [p.name for p in paths], maps


def restore_downsampled_with_sinsr(
    input_frames_dir: str,
    output_frames_dir: str,
    downscale_maps: np.ndarray,
    block_size: int,
    *,
    fp32: bool = False,
    devices: Optional[Sequence[DeviceSpec]] = None,
    parallel_chunk_length: Optional[int] = None,
    per_device_workers: int = 1,
    seed: int = 42,
    schedule: str = "staged",
    _shard_fn: Optional[ShardFn] = None,
    **model_kwargs,
) -> None:
    """Adaptive SinSR restoration over a directory of frames: drop-in for
    `restore_downsampled_with_realesrgan`.  `downscale_maps[f, by, bx]` = log2 of the block's downscale
    factor (elvis.py:2146, 2558).  `schedule="staged"` is the reference's coarse-to-fine loop
    (elvis.py:2570-2598) with 4x stages; "single4x" is the north-star single 4x call from the /4 level.
    The Real-ESRGAN keywords of the reference call (model_name, denoise_strength, tile, tile_pad, pre_pad)
    are accepted and ignored, as are `parallel_chunk_length` / `per_device_workers` (the reference ignores
    them too, elvis.py:2698-2699)."""
    _ = (parallel_chunk_length, per_device_workers)
    names, maps = _frames_and_maps(input_frames_dir, downscale_maps, "Downscale maps")
    clear_directory(output_frames_dir)
    os.makedirs(output_frames_dir, exist_ok=True)
    devs = resolve_device_list(devices, prefer_cuda=True, allow_cpu_fallback=_shard_fn is not None)
    kw = dict(fp32=fp32, seed=seed, schedule=schedule)
    kw.update({k: v for k, v in model_kwargs.items() if k in ("cfg",)})
    _run_shards(_shard_fn or _sinsr_shard, input_frames_dir, output_frames_dir, names,
                chunk_for_devices(len(names), devs), maps, block_size, 0, kw if _shard_fn is None else {})


def _restore_in_place(shard_fn: ShardFn, frames_dir: str, maps, block_size: int, devices, halo: int, kw: dict,
                      what: str, allow_cpu: bool) -> None:
    names, maps = _frames_and_maps(frames_dir, maps, what)
    if maps.size == 0 or int(np.max(maps)) <= 0:
        return   # nothing degraded: the frames stay as decoded (elvis.py:3042-3044)
    devs = resolve_device_list(devices, prefer_cuda=True, allow_cpu_fallback=allow_cpu)
    gpus = [d for d in devs if d.type == "cuda"]
    workers = gpus or devs[:1]            # the reference uses every GPU, else the first device (elvis.py:3030-3031)
    chunks = chunk_for_devices(len(names), workers)
    if halo and len(chunks) > 1:
        # in-place rewrite with a temporal halo: neighbours must read the DECODED halo frames, so restore
        # into a scratch directory first and move the files over once every worker is done
        scratch = os.path.join(frames_dir, ".elvis_restore_tmp")
        os.makedirs(scratch, exist_ok=True)
        try:
            _run_shards(shard_fn, frames_dir, scratch, names, chunks, maps, block_size, halo, kw)
            for n in names:
                os.replace(os.path.join(scratch, n), os.path.join(frames_dir, n))
        finally:
            for n in os.listdir(scratch):
                os.unlink(os.path.join(scratch, n))
            os.rmdir(scratch)
    else:
        _run_shards(shard_fn, frames_dir, frames_dir, names, chunks, maps, block_size, halo, kw)


def restore_blur_adaptive(
    input_frames_dir: str,
    blur_maps: np.ndarray,
    block_size: int,
    cfg: float = 7.0,
    creative_start: float = 1.0,
    preview_start: float = 0.0,
    seed: Optional[int] = 42,
    devices: Optional[Sequence[DeviceSpec]] = None,
    batch_size: int = 4,
    parallel_chunk_length: Optional[int] = None,
    *,
    fp32: bool = False,
    _shard_fn: Optional[ShardFn] = None,
) -> None:
    """ELVIS v2 Blur client side over a directory, IN PLACE: drop-in for `restore_with_instantir_adaptive`
    with the SwinTormer-style deblurrer in the model slot.  `blur_maps[f, by, bx]` = blur rounds
    (elvis.py:2176); per round every still-active frame is restored, finished blocks are re-pasted from
    the decoded frame, positive entries are decremented (elvis.py:2947-2981).  The diffusion keywords
    (cfg, creative_start, preview_start) and the seed have no meaning for a deterministic forward pass
    and are ignored; `parallel_chunk_length` is ignored like in the reference (elvis.py:3015)."""
    _ = (cfg, creative_start, preview_start, seed, parallel_chunk_length)
    if batch_size < 1:
        raise ValueError("`batch_size` must be at least 1.")
    kw = {} if _shard_fn is not None else dict(batch_size=batch_size, fp32=fp32)
    _restore_in_place(_shard_fn or _blur_shard, input_frames_dir, blur_maps, block_size, devices, 0, kw,
                      "blur_maps", _shard_fn is not None)


def restore_dct_adaptive(
    input_frames_dir: str,
    strength_maps: np.ndarray,
    block_size: int,
    devices: Optional[Sequence[DeviceSpec]] = None,
    *,
    fp32: bool = False,
    temporal_radius: int = 3,
    _shard_fn: Optional[ShardFn] = None,
) -> None:
    """ELVIS v2 DCT client side over a directory, IN PLACE (the build's definition of the slot; the
    reference has none): one pass of the LaplacianVCAR-style restorer, `level > 0 ? restored : decoded`.
    Each worker also reads `temporal_radius` decoded frames on either side of its range (read-only
    overlap, the expand-then-trim pattern of elvis.py:1550-1566, 1650-1657)."""
    kw = {} if _shard_fn is not None else dict(fp32=fp32)
    _restore_in_place(_shard_fn or _dct_shard, input_frames_dir, strength_maps, block_size, devices,
                      max(0, int(temporal_radius)), kw, "strength_maps", _shard_fn is not None)
