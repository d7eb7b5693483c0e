#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's OWN pure-numpy functions.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (where /root/reference
is mounted); never on the GPU box.  The reference's `elvis.py` / `utils.py`
fail to import as shipped (`ModuleNotFoundError: cv2`); following
SURVEY.md section 8c we register EMPTY `types.ModuleType` stubs for the absent
third-party names so the modules import, and then call only functions that have
no dependency on those names.  No stub implements any behaviour: anything that
touches cv2 / a model package is NOT exercised here and stays "parity unpinned"
(see DESIGN.md).

Only inputs and outputs (data) are stored - no reference source text.

    python oracle/make_golden.py            # writes tests/golden/*.npz
"""
from __future__ import annotations

import io
import os
import sys
import types
import contextlib

import numpy as np

REF = os.environ.get("ELVIS_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _stub(name: str, **attrs) -> types.ModuleType:
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def import_reference():
    """Import reference `elvis` and `utils` with empty stubs for absent packages."""
    sys.dont_write_bytecode = True
    for name in ("cv2", "pytorch_msssim"):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                _stub(name)
    try:
        import lpips  # noqa: F401
    except ImportError:
        _stub("lpips", LPIPS=object)
    try:
        import skimage.metrics  # noqa: F401
    except ImportError:
        sk = _stub("skimage")
        sk.metrics = _stub("skimage.metrics", structural_similarity=None)
    for name, attrs in (
        ("fvmd", {}),
        ("fvmd.datasets", {}),
        ("fvmd.datasets.video_datasets", {"VideoDataset": object}),
        ("fvmd.keypoint_tracking", {"track_keypoints": None}),
        ("fvmd.extract_motion_features", {"calc_hist": None}),
        ("fvmd.frechet_distance", {"calculate_fd_given_vectors": None}),
        ("instantir", {"InstantIRRuntime": object, "load_runtime": None, "restore_images_batch": None}),
    ):
        if name not in sys.modules:
            _stub(name, **attrs)
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import elvis as ref_elvis  # type: ignore
        import utils as ref_utils  # type: ignore
    return ref_elvis, ref_utils


def rng_frames(rng, n, h, w):
    return [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(n)]


def main() -> None:
    import torch

    ref_elvis, ref_utils = import_reference()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20260501)

    # ---- a1/a2: split / combine / stretch (elvis.py:1369-1455) -------------
    blocks_fx = {}
    for idx, (h, w, b) in enumerate([(16, 24, 8), (32, 32, 4), (24, 40, 8), (8, 8, 8), (64, 48, 16)]):
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        blk = ref_elvis.split_image_into_blocks(img, b)
        back = ref_elvis.combine_blocks_into_image(blk)
        blocks_fx[f"img{idx}"] = img
        blocks_fx[f"b{idx}"] = np.int64(b)
        blocks_fx[f"blocks{idx}"] = np.ascontiguousarray(blk)
        blocks_fx[f"back{idx}"] = np.ascontiguousarray(back)
    # divisibility error
    try:
        ref_elvis.split_image_into_blocks(np.zeros((10, 16, 3), np.uint8), 8)
        blocks_fx["raises_value_error"] = np.int64(0)
    except ValueError:
        blocks_fx["raises_value_error"] = np.int64(1)
    # stretch_frame (mask placement; elvis.py:1436)
    mask = (rng.random((3, 6)) < 0.34).astype(np.int8)
    # every row must keep the same count of blocks -> make it so
    mask[:] = 0
    for r in range(3):
        mask[r, rng.choice(6, size=2, replace=False)] = 1
    shrunk = rng.integers(0, 256, size=(3 * 8, 4 * 8, 3), dtype=np.uint8)
    blocks_fx["stretch_mask"] = mask
    blocks_fx["stretch_in"] = shrunk
    blocks_fx["stretch_out"] = ref_elvis.stretch_frame(shrunk, mask, 8)
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **blocks_fx)

    # ---- a9/a10: chunk_for_devices / parallel_process_frames ---------------
    chunk_fx = {}
    cases = [(30, 8, 1), (240, 8, 1), (7, 3, 1), (3, 8, 1), (0, 4, 1), (5, 1, 1), (9, 4, 3), (16, 5, 4), (1, 2, 1)]
    rows = []
    for total, ndev, mcs in cases:
        devs = [torch.device("cpu")] * ndev
        specs = ref_elvis.chunk_for_devices(total, devs, min_chunk_size=mcs)
        for s in specs:
            rows.append((total, ndev, mcs, s.start, s.end, s.chunk_id))
        if not specs:
            rows.append((total, ndev, mcs, -1, -1, -1))
    chunk_fx["chunk_table"] = np.asarray(rows, dtype=np.int64)

    frames = [np.full((2, 2, 3), i, np.uint8) for i in range(11)]

    def proc(fr, dev):
        return [f + 100 for f in fr]

    out = ref_elvis.parallel_process_frames(proc, frames, [torch.device("cpu")] * 3)
    chunk_fx["ppf_auto"] = np.stack(out)
    out = ref_elvis.parallel_process_frames(proc, frames, [torch.device("cpu")] * 2, chunk_size=4)
    chunk_fx["ppf_fixed4"] = np.stack(out)
    out = ref_elvis.parallel_process_frames(proc, frames, [torch.device("cpu")])
    chunk_fx["ppf_single"] = np.stack(out)
    chunk_fx["ppf_in"] = np.stack(frames)
    np.savez_compressed(os.path.join(OUT, "chunks.npz"), **chunk_fx)

    # ---- a13: resource_aware_restore (utils.py:176-326) ---------------------
    tiler_fx = {}

    def ident(frames, device=None, **kw):
        return [f.copy() for f in frames]

    def affine(frames, device=None, **kw):
        return [np.clip(f.astype(np.float32) * 0.5 + 7.0, 0, 255).astype(np.uint8) for f in frames]

    def coord_dep(frames, device=None, tile_coords=None, **kw):
        # depends on tile_coords so overlapping tiles disagree -> exercises the feather
        t0, t1, y0, y1, x0, x1 = tile_coords if tile_coords else (0, 0, 0, 0, 0, 0)
        add = (y0 * 3 + x0 * 5 + t0 * 11) % 37
        return [np.clip(f.astype(np.int32) + add, 0, 255).astype(np.uint8) for f in frames]

    fns = {"ident": ident, "affine": affine, "coord": coord_dep}
    cfgs = [
        # (n, h, w, tile, halo, chunk, overlap)
        (3, 64, 96, 32, 8, 0, 0),
        (3, 64, 96, 32, 0, 0, 0),
        (2, 70, 90, 32, 16, 0, 0),
        (6, 40, 40, 0, 0, 4, 2),
        (9, 48, 80, 32, 8, 4, 1),
        (2, 24, 24, 128, 16, 8, 2),  # nothing triggers -> direct call
        (2, 72, 88, 32, 16, 0, 0),
        (5, 33, 47, 16, 4, 3, 1),
    ]
    k = 0
    for (n, h, w, tile, halo, chunk, ov) in cfgs:
        fr = rng_frames(rng, n, h, w)
        for fname, fn in fns.items():
            raised = 0
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    out = ref_utils.resource_aware_restore(
                        fn, fr, tile_size=tile, halo=halo, chunk_size=chunk, chunk_overlap=ov,
                        max_workers=1, device="cpu")
            except ValueError:
                # the reference's own feather broadcast fails when an edge tile is
                # thinner than halo//2 (utils.py:287-294); recorded as behaviour.
                raised, out = 1, fr
            tiler_fx[f"c{k}_cfg"] = np.asarray([n, h, w, tile, halo, chunk, ov], np.int64)
            tiler_fx[f"c{k}_fn"] = np.asarray(fname)
            tiler_fx[f"c{k}_in"] = np.stack(fr)
            tiler_fx[f"c{k}_out"] = np.stack(out)
            tiler_fx[f"c{k}_raised"] = np.int64(raised)
            k += 1
    tiler_fx["count"] = np.int64(k)
    np.savez_compressed(os.path.join(OUT, "tiler.npz"), **tiler_fx)

    # ---- a14: adaptive_restore gate (utils.py:329-394) ----------------------
    gate_fx = {}
    maps = np.zeros((6, 8, 12), np.int32)
    maps[1, 2, 3] = 2
    maps[4, 7, 11] = 1
    maps[2, 0, 0] = 3
    calls = []

    def marker(frames, **kw):
        return [f + 1 for f in frames]

    coords = [
        (0, 2, 0, 32, 0, 32), (0, 2, 32, 64, 32, 64), (1, 3, 16, 48, 16, 64), (3, 6, 96, 128, 160, 192),
        (4, 6, 0, 16, 0, 16), (2, 3, 0, 16, 0, 16), (5, 9, 0, 128, 0, 192), (0, 6, 100, 128, 150, 192),
        (0, 1, 0, 128, 0, 192), (6, 8, 0, 16, 0, 16),
    ]
    decisions = []
    fr = [np.zeros((4, 4, 3), np.uint8)]
    for thr in (0.0, 1.0, 2.5):
        for tc in coords:
            out = ref_utils.adaptive_restore(marker, fr, degradation_maps=maps, block_size=16,
                                             tile_coords=tc, threshold=thr)
            decisions.append(list(tc) + [int(thr * 10), int(out[0][0, 0, 0])])
    out = ref_utils.adaptive_restore(marker, fr, degradation_maps=None)
    gate_fx["no_maps_restores"] = np.int64(out[0][0, 0, 0])
    out = ref_utils.adaptive_restore(marker, fr, degradation_maps=maps, block_size=16, tile_coords=None)
    gate_fx["no_coords_restores"] = np.int64(out[0][0, 0, 0])
    gate_fx["maps"] = maps
    gate_fx["decisions"] = np.asarray(decisions, np.int64)
    np.savez_compressed(os.path.join(OUT, "gate.npz"), **gate_fx)

    # ---- a17: _extract_tile_with_halo (utils.py:1227-1250) ------------------
    halo_fx = {}
    frame = rng.integers(0, 256, size=(40, 56, 3), dtype=np.uint8)
    halo_fx["frame"] = frame
    q = []
    for j, (y, x, th, tw, halo) in enumerate([(0, 0, 16, 16, 4), (24, 40, 16, 16, 8), (8, 8, 16, 24, 0), (30, 50, 16, 16, 16)]):
        tile, bounds = ref_utils._extract_tile_with_halo(frame, y, x, th, tw, halo)
        halo_fx[f"tile{j}"] = tile
        q.append([y, x, th, tw, halo] + list(bounds))
    halo_fx["queries"] = np.asarray(q, np.int64)
    np.savez_compressed(os.path.join(OUT, "halo.npz"), **halo_fx)

    # ---- PSNR / MSE definitions (elvis.py:627-671) --------------------------
    psnr_fx = {}
    a = rng.integers(0, 256, size=(5, 32, 48, 3), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-6, 7, size=a.shape), 0, 255).astype(np.uint8)
    b[3] = a[3]  # identical frame -> 100 dB cap
    m = rng.random((5, 32, 48)) < 0.4
    m[4] = False  # empty mask -> 100 dB
    psnr_fx["a"], psnr_fx["b"], psnr_fx["mask"] = a, b, m
    psnr_fx["psnr_full"] = np.asarray([ref_elvis._masked_psnr(a[i], b[i]) for i in range(5)], np.float64)
    psnr_fx["psnr_masked"] = np.asarray([ref_elvis._masked_psnr(a[i], b[i], m[i]) for i in range(5)], np.float64)
    psnr_fx["mse_full"] = np.asarray([ref_elvis._masked_mse(a[i], b[i]) for i in range(5)], np.float64)
    psnr_fx["mse_masked"] = np.asarray([ref_elvis._masked_mse(a[i], b[i], m[i]) for i in range(5)], np.float64)
    np.savez_compressed(os.path.join(OUT, "psnr.npz"), **psnr_fx)

    # ---- strength-map npz wire format (elvis.py:2247-2272) ------------------
    import tempfile
    maps = rng.integers(0, 4, size=(4, 6, 10)).astype(np.int32)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "maps.npz")
        with contextlib.redirect_stdout(io.StringIO()):
            ref_elvis.encode_strength_maps_to_npz(maps, p)
            dec = ref_elvis.decode_strength_maps_from_npz(p)
        raw = open(p, "rb").read()
    np.savez_compressed(os.path.join(OUT, "strength_maps.npz"), maps_in=maps, maps_decoded=dec,
                        wire_bytes=np.frombuffer(raw, np.uint8))

    # ---- a18: _resolve_device_list on a CPU-only host (elvis.py:451-530) ----
    res = {}
    if not torch.cuda.is_available():
        res["default"] = np.asarray([str(d) for d in ref_elvis._resolve_device_list(None)])
        res["cpu_dup"] = np.asarray([str(d) for d in ref_elvis._resolve_device_list(["cpu", "cpu", torch.device("cpu")])])
        errs = []
        for spec in ([0], ["cuda"], ["cuda:1"]):
            try:
                ref_elvis._resolve_device_list(spec)
                errs.append(0)
            except ValueError:
                errs.append(1)
        res["cuda_specs_raise"] = np.asarray(errs, np.int64)
        try:
            ref_elvis._resolve_device_list(None, allow_cpu_fallback=False)
            res["no_fallback_raises"] = np.int64(0)
        except ValueError:
            res["no_fallback_raises"] = np.int64(1)
        np.savez_compressed(os.path.join(OUT, "devices_cpu.npz"), **res)

    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print("  ", f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
