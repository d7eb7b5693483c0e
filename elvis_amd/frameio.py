"""Frame and side-channel file formats on either side of the restoration hot path (SURVEY.md 8f row f1
and the IO helpers the directory drivers need).

* PNG / JPEG frames as BGR uint8 HWC arrays - what `cv2.imread(..., IMREAD_COLOR)` / `cv2.imwrite`
  give the reference (elvis.py:123-136); read and written through PIL here (cv2 is not a dependency).
* strength maps: `np.savez_compressed(path, strength_maps=uint8[F,By,Bx])` (elvis.py:2247-2272).
* block masks: `np.packbits` of a uint8 {0,1} array + its shape (elvis.py:4412-4418, 4537-4539).

Host code only: these are wire formats, not arithmetic.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import List, Sequence, Union

import numpy as np

FRAME_SUFFIXES = (".png", ".jpg", ".jpeg")
PathLike = Union[str, os.PathLike]


def get_frame_paths(directory: PathLike) -> List[Path]:
    """Image files of `directory`, sorted by name; [] when it is not a directory (elvis.py:235-241)."""
    d = Path(directory)
    if not d.is_dir():
        return []
    return sorted(p for p in d.iterdir() if p.suffix.lower() in FRAME_SUFFIXES)


def clear_directory(directory: PathLike, patterns: Sequence[str] = ("*.png", "*.jpg", "*.jpeg")) -> None:
    """Delete the files matching `patterns`; a missing directory is not an error (elvis.py:223-232)."""
    d = Path(directory)
    if d.is_dir():
        for pattern in patterns:
            for p in d.glob(pattern):
                if p.is_file():
                    p.unlink()


def load_frame(path: PathLike) -> np.ndarray:
    """One frame as a BGR uint8 (H,W,3) array; IOError when it cannot be read (elvis.py:123-128)."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    except (OSError, ValueError) as exc:
        raise IOError(f"Failed to load frame: {path}") from exc
    return np.ascontiguousarray(rgb[:, :, ::-1])


def save_frame(frame: np.ndarray, path: PathLike) -> None:
    """Write a BGR uint8 (H,W,3) frame (the format follows the file suffix), creating the directory
    (elvis.py:131-135)."""
    from PIL import Image
    frame = np.asarray(frame)
    if frame.dtype != np.uint8 or frame.ndim != 3 or frame.shape[2] != 3:
        raise IOError(f"Failed to save frame: {path} (expected uint8 HxWx3, got {frame.dtype} {frame.shape})")
    os.makedirs(os.path.dirname(os.fspath(path)) or ".", exist_ok=True)
    try:
        Image.fromarray(np.ascontiguousarray(frame[:, :, ::-1]), "RGB").save(path)
    except (OSError, ValueError) as exc:
        raise IOError(f"Failed to save frame: {path}") from exc


def load_frames(directory: PathLike) -> List[np.ndarray]:
    return [load_frame(p) for p in get_frame_paths(directory)]


# ----------------------------------------------------------------------------- strength maps (f1)
def encode_strength_maps_to_npz(strength_maps, output_path: PathLike) -> None:
    """uint8 [F,By,Bx] under the key `strength_maps`, zlib-compressed (elvis.py:2247-2259).  Values are
    cast to uint8 like the reference does (levels are 0..10)."""
    maps = np.stack(strength_maps, axis=0) if isinstance(strength_maps, (list, tuple)) else np.asarray(strength_maps)
    np.savez_compressed(output_path, strength_maps=maps.astype(np.uint8, copy=False))


def decode_strength_maps_from_npz(npz_path: PathLike) -> np.ndarray:
    """The [F,By,Bx] array stored by `encode_strength_maps_to_npz`; FileNotFoundError when the file is
    missing (elvis.py:2261-2272).  Loaded with allow_pickle=False."""
    if not os.path.exists(npz_path):
        raise FileNotFoundError(f"Strength maps file not found: {npz_path}")
    with np.load(npz_path, allow_pickle=False) as data:
        return np.array(data["strength_maps"])


load_strength_maps = decode_strength_maps_from_npz


# ----------------------------------------------------------------------------- packed block masks
def save_block_masks(masks, path: PathLike) -> None:
    """{0,1} block masks of any shape as `packed` (np.packbits of the flattened uint8 array) + `shape`
    (elvis.py:4412-4418)."""
    m = np.asarray(masks, dtype=np.uint8)
    np.savez(path, packed=np.packbits(m), shape=m.shape)


def load_block_masks(path: PathLike) -> np.ndarray:
    """Inverse of `save_block_masks`: unpack, drop the bit padding, restore the shape (elvis.py:4537-4539)."""
    with np.load(path, allow_pickle=False) as data:
        shape = tuple(int(v) for v in data["shape"])
        bits = np.unpackbits(data["packed"])
    return bits[: int(np.prod(shape))].reshape(shape)
