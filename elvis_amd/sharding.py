"""Frame-parallel sharding - the reference's P4 surface, host logic only.

Interface kept from the reference so the pipeline can import these names unchanged:
`ChunkSpec` (elvis.py:246), `chunk_for_devices` (elvis.py:255), `parallel_process_frames`
(elvis.py:283), `_resolve_device_list` (elvis.py:451).  The bodies are written from that interface and
from the golden tables in tests/golden/chunks.npz / devices_cpu.npz, around one primitive: the
even split `rank_frame_range`, which is also what the one-process-per-GPU runner
(`elvis_amd.distributed`) and the directory drivers (`elvis_amd.drivers`) use.
"""
from __future__ import annotations

import re
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Callable, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

DeviceSpec = Union[int, str, torch.device]


@dataclass
class ChunkSpec:
    """A contiguous frame range [start, end) assigned to `device`; `chunk_id` orders the results."""
    start: int
    end: int
    device: torch.device
    chunk_id: int = 0


def rank_frame_range(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[start, end) of part `rank` when `total` frames are dealt to `world_size` parts as evenly as
    possible, larger parts first (the rule of elvis.py:264-271 and of `_split_ranges`,
    elvis.py:3046-3060)."""
    base, extra = divmod(max(total, 0), world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def chunk_for_devices(total: int, devices: List[torch.device], min_chunk_size: int = 1) -> List[ChunkSpec]:
    """One contiguous chunk per device (even split, larger chunks first); chunk_id = device position.
    A device after the first whose share is below `min_chunk_size` gets no chunk, and - as in the
    reference, pinned by the golden table - its share is not handed to anyone else: later chunks
    continue where the last kept chunk ended."""
    if total <= 0 or not devices:
        return []
    world = len(devices)
    chunks: List[ChunkSpec] = []
    cursor = 0
    for cid, dev in enumerate(devices):
        lo, hi = rank_frame_range(total, world, cid)
        size = hi - lo
        if size > 0 and (cid == 0 or size >= min_chunk_size):
            chunks.append(ChunkSpec(cursor, cursor + size, dev, cid))
            cursor += size
    return chunks


def _fixed_size_chunks(total: int, devices: Sequence[torch.device], chunk_size: int) -> List[ChunkSpec]:
    """Chunks of `chunk_size` frames (the last one shorter) dealt round-robin to the devices."""
    starts = range(0, total, chunk_size)
    return [ChunkSpec(s, min(s + chunk_size, total), devices[cid % len(devices)], cid) for cid, s in enumerate(starts)]


def parallel_process_frames(
    process_fn: Callable[[List[np.ndarray], torch.device], List[np.ndarray]],
    frames: List[np.ndarray],
    devices: List[torch.device],
    chunk_size: Optional[int] = None,
    max_workers: Optional[int] = None,
) -> List[np.ndarray]:
    """Run `process_fn(frames[start:end], device)` for every chunk on a thread pool and concatenate the
    results in chunk order; a single chunk runs inline.  `process_fn` must be thread-safe per device
    and must select its device itself (pool threads start on device 0)."""
    if not frames:
        return []
    devices = list(devices) or [torch.device("cpu")]
    plan = (chunk_for_devices(len(frames), devices) if chunk_size is None
            else _fixed_size_chunks(len(frames), devices, chunk_size))
    if not plan:
        return []

    def run(c: ChunkSpec) -> List[np.ndarray]:
        return process_fn(frames[c.start:c.end], c.device)

    if len(plan) == 1:
        return run(plan[0])
    with ThreadPoolExecutor(max_workers=max_workers or min(len(plan), len(devices))) as pool:
        parts = list(pool.map(run, plan))   # map() yields in submission (= chunk_id) order
    return [frame for part in parts for frame in part]


# ----------------------------------------------------------------------------- device specs
_CUDA_SPEC = re.compile(r"^cuda(?::(?P<index>.*))?$")


def _parse_device(spec: DeviceSpec, gpu_count: int) -> torch.device:
    """One specifier -> torch.device, validated against the visible GPU count.  Accepted forms:
    int n | "cuda" | "cuda:n" | torch.device | any other torch device string ("cpu", ...)."""
    if isinstance(spec, torch.device):
        dev = spec
    elif isinstance(spec, int) and not isinstance(spec, bool):
        if gpu_count == 0:
            raise ValueError(f"device index {spec} given but no GPU is visible")
        dev = torch.device("cuda", spec) if spec >= 0 else None
        if dev is None:
            raise ValueError(f"device index {spec} is negative")
    else:
        text = str(spec)
        m = _CUDA_SPEC.match(text)
        if m is None:
            dev = torch.device(text)
        else:
            if gpu_count == 0:
                raise ValueError(f"'{text}' requested but no GPU is visible")
            idx_text = m.group("index")
            if not idx_text:                      # "cuda" / "cuda:" -> the current device
                dev = torch.device("cuda")
            elif idx_text.isdigit():
                dev = torch.device("cuda", int(idx_text))
            else:
                raise ValueError(f"malformed device string '{text}'")
    if dev.type == "cuda" and (dev.index or 0) >= gpu_count:
        raise ValueError(f"{dev} is not available: {gpu_count} GPU(s) visible")
    return dev


def _unique(devs: Iterable[torch.device]) -> List[torch.device]:
    seen, out = set(), []
    for d in devs:
        key = (d.type, (d.index or 0) if d.type == "cuda" else d.index)   # "cuda" and "cuda:0" are one device
        if key not in seen:
            seen.add(key)
            out.append(d)
    return out


def resolve_device_list(
    devices: Optional[Sequence[DeviceSpec]],
    *,
    prefer_cuda: bool = True,
    allow_cpu_fallback: bool = True,
) -> List[torch.device]:
    """Normalise device specifiers into a list of distinct torch.device entries, first occurrence
    first.  No specifiers: every visible GPU ("cuda:N" is what PyTorch-ROCm calls the MI355X devices),
    else the CPU if `allow_cpu_fallback`.  Anything unusable raises ValueError."""
    gpu_count = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if devices:
        resolved = _unique(_parse_device(s, gpu_count) for s in devices)
    elif prefer_cuda and gpu_count:
        resolved = [torch.device("cuda", i) for i in range(gpu_count)]
    else:
        resolved = []
    if resolved:
        return resolved
    if not allow_cpu_fallback:
        raise ValueError("no usable compute device and the CPU fallback is disabled")
    return [torch.device("cpu")]


_resolve_device_list = resolve_device_list  # the reference's private name (elvis.py:451)
