"""Frame-parallel sharding helpers - the reference's P4 surface, host logic only.

`ChunkSpec`, `chunk_for_devices`, `parallel_process_frames` and `resolve_device_list` keep the
names, argument meaning and error behaviour of elvis.py:246-353 and elvis.py:451-530 so the
existing pipeline can import them from here unchanged.  `rank_frame_range` applies the same
split rule to torch.distributed ranks (one process per GPU).
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor, as_completed
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch


@dataclass
class ChunkSpec:
    """Specification for a processing chunk (elvis.py:246-252)."""
    start: int
    end: int
    device: torch.device
    chunk_id: int = 0


def chunk_for_devices(total: int, devices: List[torch.device], min_chunk_size: int = 1) -> List[ChunkSpec]:
    """Contiguous split, one chunk per device; the first `total % D` devices get one extra
    frame; empty chunks are dropped (elvis.py:255-280)."""
    if not devices or total <= 0:
        return []
    num = len(devices)
    base, rem = divmod(total, num)
    chunks: List[ChunkSpec] = []
    start = 0
    for idx, device in enumerate(devices):
        size = base + (1 if idx < rem else 0)
        if size < min_chunk_size and idx > 0:
            continue  # (the reference skips the chunk without advancing `start`)
        end = start + size
        if end > start:
            chunks.append(ChunkSpec(start=start, end=end, device=device, chunk_id=idx))
        start = end
    return chunks


def parallel_process_frames(
    process_fn: Callable[[List[np.ndarray], torch.device], List[np.ndarray]],
    frames: List[np.ndarray],
    devices: List[torch.device],
    chunk_size: Optional[int] = None,
    max_workers: Optional[int] = None,
) -> List[np.ndarray]:
    """Thread-pool map of `process_fn(frames_chunk, device)` over chunks, results reassembled in
    ascending chunk_id order (elvis.py:283-353).  `process_fn` must be thread-safe per device."""
    if not frames:
        return []
    if not devices:
        devices = [torch.device("cpu")]
    n = len(frames)
    if chunk_size is None:
        chunks = chunk_for_devices(n, devices)
    else:
        chunks, cursor, cid = [], 0, 0
        while cursor < n:
            end = min(cursor + chunk_size, n)
            chunks.append(ChunkSpec(start=cursor, end=end, device=devices[cid % len(devices)], chunk_id=cid))
            cursor, cid = end, cid + 1
    if not chunks:
        return []
    if len(chunks) == 1:
        c = chunks[0]
        return process_fn(frames[c.start:c.end], c.device)
    results: Dict[int, List[np.ndarray]] = {}
    workers = max_workers or min(len(chunks), len(devices))

    def _run(chunk: ChunkSpec) -> Tuple[int, List[np.ndarray]]:
        return chunk.chunk_id, process_fn(frames[chunk.start:chunk.end], chunk.device)

    with ThreadPoolExecutor(max_workers=workers) as ex:
        futures = {ex.submit(_run, c): c for c in chunks}
        for fut in as_completed(futures):
            cid, out = fut.result()
            results[cid] = out
    output: List[np.ndarray] = []
    for c in sorted(chunks, key=lambda c: c.chunk_id):
        output.extend(results[c.chunk_id])
    return output


def resolve_device_list(
    devices: Optional[Sequence[Union[int, str, torch.device]]],
    *,
    prefer_cuda: bool = True,
    allow_cpu_fallback: bool = True,
) -> List[torch.device]:
    """Normalise device specifiers into unique torch.device entries (elvis.py:451-530).
    "cuda:N" is what PyTorch-ROCm calls the MI355X devices too."""
    gpu_count = torch.cuda.device_count() if torch.cuda.is_available() else 0

    def norm(spec) -> torch.device:
        if isinstance(spec, torch.device):
            dev = spec
        elif isinstance(spec, int):
            if not torch.cuda.is_available():
                raise ValueError("CUDA device indices were provided but no CUDA devices are available.")
            if spec < 0 or spec >= gpu_count:
                raise ValueError(f"Requested CUDA device index {spec} is out of range.")
            dev = torch.device(f"cuda:{spec}")
        else:
            s = str(spec)
            if s.startswith("cuda"):
                if not torch.cuda.is_available():
                    raise ValueError("CUDA devices were requested but CUDA is not available.")
                if s in ("cuda", "cuda:"):
                    dev = torch.device("cuda")
                else:
                    try:
                        idx = int(s.split(":", 1)[1])
                    except (IndexError, ValueError):
                        raise ValueError(f"Invalid CUDA device string '{s}'.") from None
                    if idx < 0 or idx >= gpu_count:
                        raise ValueError(f"Requested CUDA device {s} exceeds detected count {gpu_count}.")
                    dev = torch.device(f"cuda:{idx}")
            else:
                dev = torch.device(s)
        if dev.type == "cuda":
            idx = dev.index if dev.index is not None else 0
            if idx < 0 or idx >= gpu_count:
                raise ValueError(f"Requested CUDA device {idx} is not available. Detected {gpu_count} device(s).")
        return dev

    if not devices:
        if prefer_cuda and gpu_count > 0:
            specs: Sequence = [f"cuda:{i}" for i in range(gpu_count)]
        elif allow_cpu_fallback:
            specs = ["cpu"]
        else:
            raise ValueError("No CUDA devices available and CPU fallback disabled.")
    else:
        specs = devices
    out: List[torch.device] = []
    seen = set()
    for spec in specs:
        dev = norm(spec)
        key = str(dev)
        if dev.type == "cuda":
            key = f"cuda:{dev.index if dev.index is not None else 0}"
        if key in seen:
            continue
        seen.add(key)
        out.append(dev)
    if not out:
        if allow_cpu_fallback:
            out.append(torch.device("cpu"))
        else:
            raise ValueError("No valid compute devices resolved from the provided specification.")
    return out


_resolve_device_list = resolve_device_list  # the reference's private name (elvis.py:451)


def rank_frame_range(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[start,end) of the frames rank `rank` owns under the chunk_for_devices rule
    (elvis.py:264-271; same rule as _split_ranges, elvis.py:3046-3060)."""
    base, rem = divmod(total, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
