#!/usr/bin/env python3
"""Benchmark of the ELVIS v2 client-side restore hot path on MI355X (headline: Downsample / SinSR 4x).

    python bench.py --gpus N --steps K --warmup W            # one JSON line on stdout (rank 0)

`--gpus N` with N > 1 and no torchrun environment starts the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N ... bench.py ...`, before this process touches the
GPU); under torchrun (RANK / WORLD_SIZE set) it runs as one rank.  One process per GPU, RCCL over xGMI.

A "step" is one pass of the hot path over one clip per rank, timed HOST TO HOST as SURVEY.md 8(d) defines the
metric: `--frames` (30) degraded 1080p uint8 frames + block maps in pinned host memory -> upload -> /4 area
downscale -> SinSR 4x (VQ-f4 encode, Swin-UNet single step, VQ-f4 decode; sampler noise generated per step on
the host's cores, keyed on the global frame index) -> block-map recompose -> (N > 1: ONE all-gather of the
restored clip over RCCL) -> restored uint8 frames in pinned host memory (rank 0 holds the whole sequence).
Transfers and the noise generator overlap compute on side streams (restore.restore_clip_single4x_host).
Weak scaling: every rank owns its own 30-frame clip (the chunk_for_devices split of a 30*N-frame clip,
elvis.py:255-280).  value = frames restored by all ranks / max-over-ranks step time.

Extra keys of the JSON line
  hbm_resident - the same step with inputs and outputs resident in HBM (no PCIe, no noise generation): the
                 round-1 figure, kept for comparison; never `value`
  roofline     - the dominant kernel (implicit-GEMM conv on MFMA): algorithmic FLOPs / HIP-event time measured
                 live on the launch stream during the hbm_resident steps, vs the dense f16 MFMA peak
                 (MI355X_MICROARCH.md: ~2.5 PFLOP/s; fp32 MFMA 157.3 TFLOP/s)
  cpu_baseline - the CPU oracle (oracle/sinsr_ref.py, PyTorch fp32) timed on this host's cores on one 768x768
                 output region, scaled to 1080p frames/s
  parity       - max-abs / PSNR / VQ code agreement of the benchmarked mode against the CPU oracle on a 256x256 tile
  in_tolerance - the precision modes that meet the north star's tolerance (x3, dec_f16), each measured exactly like
                 `value` (30 frames host to host, warm, 3 steps) with its own roofline and parity
  staged_schedule - the drop-in default schedule (reference loop elvis.py:2570-2598) through restore_frames_sinsr
  slots        - BASELINE configs 3 and 4 (ELVIS v2 DCT / Blur): 30 frames host to host, 3 steps, kernel table, parity
  dist         - (N > 1 or --force-dist) backend, world size and the check that the gathered sequence holds rank 0's shard
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3}
HBM_PEAK_GBS = 8000.0
TRAFFIC_FILE = "profiles/r03_hbm_traffic.json"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=30, help="frames per rank per step")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--block", type=int, default=8)
    ap.add_argument("--mode", choices=["f16", "f32"], default="f16")
    ap.add_argument("--slot", choices=["downsample", "dct", "blur"], default="downsample",
                    help="which ELVIS v2 client path is the timed workload (BASELINE configs 2 / 3 / 4)")
    ap.add_argument("--no-fuse-gn", dest="fuse_gn", action="store_false",
                    help="run GroupNorm-apply+SiLU as separate elementwise passes instead of the conv prologue")
    ap.add_argument("--batch", type=int, default=15, help="frames per network invocation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-slots", action="store_true", help="skip the short DCT / Blur slot runs")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend of the N > 1 run: nccl (= RCCL over xGMI, the measurement) or gloo (rehearsal of "
                         "the multi-rank path on a box with fewer GPUs than ranks: ranks then share devices)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the distributed branch even at world size 1: init_process_group (RCCL), the all-gather at world 1 and "
                         "rank 0's download of the gathered clip - exercises the N > 1 code path on a one-GPU box")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the in_tolerance / staged / slots blocks (the headline, roofline, cpu_baseline and parity stay)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU check of the --gpus launcher: gloo ranks, one all-gather of synthetic frames, no kernels")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------- launcher (before any GPU call)
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks_if_needed(args) -> bool:
    """`--gpus N` (N > 1) outside a torchrun environment: start N ranks as a CHILD process tree and exit with
    its code.  Nothing here initialises the GPU (no torch.cuda call), and nothing is exec'ed."""
    if (args.gpus <= 1 and not args.force_dist) or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return False
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={max(1, args.gpus)}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def launcher_selftest(args):
    """What a rank does around the timed region, without kernels: rendezvous (gloo), shard bookkeeping, the
    single all-gather, the max-over-ranks reduction and rank 0's JSON line.  Not a measurement."""
    import torch
    import torch.distributed as dist
    from elvis_amd.distributed import all_gather_frames
    from elvis_amd.sharding import rank_frame_range
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    use_dist = world > 1 or (args.force_dist and "RANK" in os.environ)
    if use_dist:
        dist.init_process_group("gloo")
    total = world * 3 + 1                                     # ragged shards
    s, e = rank_frame_range(total, world, rank)
    local = torch.arange(s, e, dtype=torch.uint8).view(-1, 1, 1, 1).expand(-1, 2, 2, 3).contiguous()
    t0 = time.perf_counter()
    full = all_gather_frames(local, total) if use_dist else local
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ok = bool((full[:, 0, 0, 0] == torch.arange(total, dtype=torch.uint8)).all())
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "requested_gpus": args.gpus, "frames": total,
                          "order_ok": ok, "backend": "gloo" if use_dist else None}))
    if use_dist:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


# ----------------------------------------------------------------------------- helpers
def host_cores() -> int:
    """CPU share actually available to this process (affinity mask and cgroup quota), not the
    machine's core count: the GPU box gives one GPU's job ~16 of its cores."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("ELVIS_CPU_THREADS", "16"))))


def cpu_baseline_and_parity(model_mode, dev, fuse_gn, extra_modes=("x3", "dec_f16")):
    """CPU baseline: the oracle timed on a 768x768 output region of the full-width config (scaled to
    1080p frames/s by pixel count).  Parity: GPU vs oracle on a 256x256 output tile (64x64 LR), for the
    benchmarked mode and for the in-tolerance modes, with and without the VQ lookup."""
    import dataclasses
    import numpy as np
    import torch
    from elvis_amd import restore
    from elvis_amd.synth import synth_clip
    from elvis_amd.weights import SinSRConfig, frame_noise, make_sinsr_weights
    from oracle import sinsr_ref as R

    cfg = SinSRConfig()
    sd = make_sinsr_weights(cfg, 0)
    lr = torch.from_numpy(synth_clip(20260501, 1, 64, 64)[0])
    noise = frame_noise(cfg, 42, 0, 64, 64)
    cores = host_cores()
    torch.set_num_threads(cores)
    R.ae_encode(sd, cfg, torch.zeros(1, 3, 32, 32))  # warm the thread pool / oneDNN primitives
    S = 192   # timed sample: a 768x768 output region (192x192 LR, 9 tiles' worth, ~10-20 s of CPU work)
    lr_big = torch.from_numpy(synth_clip(20260502, 1, S, S)[0])
    noise_big = frame_noise(cfg, 42, 0, S, S)
    t0 = time.perf_counter()
    R.sinsr_forward(sd, cfg, lr_big, noise_big)
    t_cpu = time.perf_counter() - t0
    del lr_big, noise_big
    per_frame = (1080 * 1920) / float(4 * S * 4 * S)
    base = {"value": 1.0 / (t_cpu * per_frame), "unit": "1080p frames/s (scaled by pixel count from the sample)",
            "cores": cores, "kind": "port",
            "sample": f"one {4 * S}x{4 * S} output region ({S}x{S} LR) of the full-width SinSR config through the "
                      f"CPU oracle, fp32, {t_cpu:.2f} s on {cores} threads; a 1080p frame = {per_frame:.3f} such regions"}

    cfg_c = dataclasses.replace(cfg, quantize=False)
    ref_c = R.sinsr_forward(sd, cfg_c, lr, noise)
    ref_q, st_q = R.sinsr_forward(sd, cfg, lr, noise, return_stages=True)
    _, ref_idx = R.vq_quantize(sd, st_q["z0"])

    def compare(model, quantize):
        """The cached product model of a mode (weights seed 0 = `sd`) on the parity tile; `quantize` picks the path
        (the flag is read at run time by `decode`; everything else of the config is identical)."""
        keep = model.cfg
        model.cfg = cfg if quantize else cfg_c
        try:
            st = {}
            u8, f32 = model.forward(lr[None].to(dev), noise.to(dev), want_f32=True, stages=st)
            u8b, _ = model.forward(lr[None].to(dev), noise.to(dev), want_f32=True)
            idx = model.decode(st["z0"], want_idx=True)[1] if quantize else None
        finally:
            model.cfg = keep
        ref_img = ref_q if quantize else ref_c
        ref_u8 = R.to_u8(ref_img).numpy().astype(np.float32)
        got = u8[0].cpu().numpy().astype(np.float32)
        mse = float(np.mean((got - ref_u8) ** 2))
        # north star: "PSNR within 0.01 dB" - PSNR of either output against one fixed target (the LR tile, nearest x4)
        target = np.kron(lr.numpy().astype(np.float32), np.ones((4, 4, 1), np.float32))
        psnr_t = lambda img: float(10 * np.log10(255.0 ** 2 / max(1e-12, float(np.mean((img - target) ** 2)))))
        out = {"max_abs_f32": float((f32[0].cpu() - ref_img).abs().max()),
               "psnr_db_vs_oracle_u8": float("inf") if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse)),
               "psnr_delta_db_vs_fixed_target": abs(psnr_t(got) - psnr_t(ref_u8)),
               "max_abs_u8": float(np.abs(got - ref_u8).max()),
               "bit_reproducible": bool(torch.equal(u8, u8b))}
        if quantize:
            out["vq_code_agreement"] = float((idx.cpu().long() == ref_idx).float().mean())
        return out

    def mode_parity(precision):
        m = restore.get_sinsr_model(dev, cfg=cfg, fp32=(model_mode == "f32"), fuse_gn=fuse_gn, precision=precision)
        return {"continuous_path": compare(m, False), "with_vq_lookup": compare(m, True)}

    # continuous path (no VQ lookup) is the stable parity figure; with the (discontinuous) lookup
    # in the loop a code flip anywhere shows up as a local O(0.1) difference
    parity = {"tile": "256x256 output, full-width config, weights seed 0, noise seed 42", "bar_max_abs_f32": 1e-3,
              **mode_parity(None),
              "note": "f16 operands alone give 1.6e-3 on this tile (tools/precision_study.py, DESIGN.md 4.1): no plain f16 "
                      "mode can meet 1e-3.  In-tolerance modes and their parity on this tile: the `in_tolerance` block"}
    modes = {m: mode_parity(m) for m in extra_modes} if model_mode == "f16" else {}
    torch.cuda.empty_cache()
    return base, parity, modes


def _aggregate(prof):
    agg = {}
    for name, work, e0, e1 in prof:
        a = agg.setdefault(name, [0.0, 0.0, 0])
        a[0] += work
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
    return agg


def _kernel_table(conv_prof, other_prof, el, mode, conv_mfma_factor=None):
    """Per-kernel roofline records from the live HIP events of one profiled step (`el` seconds), by time."""
    cands = []
    for name, (work, sec, cnt) in _aggregate(conv_prof).items():
        rec = {"kernel": name, "bound": "mfma", "achieved": work / sec / 1e12, "peak": PEAK_TFLOPS[mode], "unit": "TFLOP/s",
               "frac": work / sec / 1e12 / PEAK_TFLOPS[mode], "launches": cnt, "avg_launch_ms": sec / cnt * 1e3,
               "time_share": sec / el}
        if conv_mfma_factor and conv_mfma_factor(name) > 1:
            # the compensated kernels run 3 (planar) or 4 (interleaved) f16 MFMAs per algorithmic one
            rec["mfma_per_algorithmic"] = conv_mfma_factor(name)
            rec["executed_mfma_tflops"] = rec["achieved"] * conv_mfma_factor(name)
            rec["executed_mfma_frac"] = rec["frac"] * conv_mfma_factor(name)
        cands.append((sec, rec))
    for (name, bound), (work, sec, cnt) in _aggregate([((n, b), w, e0, e1) for n, b, w, e0, e1 in other_prof]).items():
        peak, unit, scale = (PEAK_TFLOPS[mode], "TFLOP/s", 1e12) if bound == "mfma" else (HBM_PEAK_GBS, "GB/s", 1e9)
        cands.append((sec, {"kernel": name, "bound": bound, "achieved": work / sec / scale, "peak": peak, "unit": unit,
                            "frac": work / sec / scale / peak, "launches": cnt, "avg_launch_ms": sec / cnt * 1e3,
                            "time_share": sec / el}))
    cands.sort(key=lambda c: -c[0])
    return [c[1] for c in cands]


def _timed_steps(step, warmup, steps):
    import torch
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def slot_parity(kind, dev):
    """f16 (the slot's default) and fp32 against oracle/restorers_ref.py on a small clip of the FULL slot config."""
    import numpy as np
    import torch
    from elvis_amd.restorers import DCNRestorer, SwinDeblur
    from elvis_amd.weights import (DCNRestorerConfig, SwinDeblurConfig, make_dcn_weights, make_deblur_weights)
    from oracle import restorers_ref as R
    rng = np.random.default_rng(3 if kind == "dct" else 4)
    out = {}
    if kind == "dct":
        cfg = DCNRestorerConfig()
        sd = make_dcn_weights(cfg, 0)
        frames = torch.from_numpy(rng.integers(0, 256, size=(5, 64, 96, 3), dtype=np.uint8))
        ref, _ = R.dcn_restore_frames(sd, cfg, frames)
        for name, dt in (("f16", torch.float16), ("f32", torch.float32)):
            got = DCNRestorer(cfg, sd, dev, dt).restore(frames.to(dev), chunk=2).cpu().numpy().astype(int)
            d = np.abs(got - ref.numpy().astype(int))
            out[name] = {"max_abs_u8": int(d.max()), "frac_differing": float((d > 0).mean())}
        out["tile"] = "5 frames 64x96, full DCN restorer config, weights seed 0; u8 frames vs oracle/restorers_ref.py (fp32 CPU)"
    else:
        cfg = SwinDeblurConfig()
        sd = make_deblur_weights(cfg, 0)
        frames = torch.from_numpy(rng.integers(0, 256, size=(2, 64, 128, 3), dtype=np.uint8))
        ref_u8, ref = R.deblur_restore_frames(sd, cfg, frames)
        for name, dt in (("f16", torch.float16), ("f32", torch.float32)):
            u8, f32 = SwinDeblur(cfg, sd, dev, dt).restore(frames.to(dev), swap_rb=False, want_f32=True)
            out[name] = {"max_abs_f32": float((f32.cpu() - ref).abs().max()),
                         "max_abs_u8": int(np.abs(u8.cpu().numpy().astype(int) - ref_u8.numpy().astype(int)).max())}
        out["tile"] = "2 frames 64x128, full Swin deblur config, weights seed 0; pre-quantisation image and u8 frames vs oracle/restorers_ref.py"
    out["oracle"] = "parity unpinned (the reference holds no source, checkpoint or fixture for this slot; DESIGN.md 2)"
    return out


def slot_run(kind, dev, mode, frames=30, steps=3, warmup=1, parity=True):
    """BASELINE config 3 (DCT: LaplacianVCAR-style DCNv2 restorer) or 4 (Blur: SwinTormer-style deblur, one round) as
    BASELINE states them: a `frames`-frame 1080p clip, HOST TO HOST (pinned host frames + maps -> upload -> restore ->
    block-map recompose -> download; restore.restore_clip_slot_host), `steps` timed steps after `warmup`.  The kernel
    table comes from one further, HBM-resident, profiled pass over the clip (HIP events on the launch stream): conv
    kernels against the MFMA peak, window attention and the DCNv2 gather against the HBM peak (algorithmic bytes)."""
    import numpy as np
    import torch
    from elvis_amd import ops, restore, synth
    H, W, B = 1080, 1920, 8
    clip = synth.synth_clip(synth.CLIP_SEED, 2, H, W)
    frames_h = torch.from_numpy(np.concatenate([clip] * ((frames + 1) // 2))[:frames]).pin_memory()
    lv = synth.synth_level_maps(synth.MAP_SEED, frames, H // B, W // B)
    lv = lv if kind == "dct" else np.minimum(lv, 1)                     # Blur: rounds = 1
    maps_h = torch.from_numpy(lv.astype(np.int32)).pin_memory()
    out_h = torch.empty_like(frames_h).pin_memory()
    net = restore._get_restorer(kind, dev, mode == "f32")
    step = lambda: restore.restore_clip_slot_host(kind, net, frames_h, maps_h, B, out_h, batch_size=2, upload_chunk=6)
    sec = _timed_steps(step, warmup, steps)
    conv_prof, other_prof = [], []
    ops.CONV_PROFILER, ops.KERNEL_PROFILER = conv_prof, other_prof
    el = _timed_steps(step, 0, 1)
    ops.CONV_PROFILER = ops.KERNEL_PROFILER = None
    table = _kernel_table(conv_prof, other_prof, el, mode)
    res = {"workload": f"{frames}-frame 1920x1080 clip, ELVIS v2 {'DCT (DCNv2 restorer)' if kind == 'dct' else 'Blur (Swin deblur, 1 round)'}, "
                       f"block {B}, host to host (pinned frames+maps -> H2D, restore, recompose, D2H)",
           "frames_per_s": frames / sec, "steps": steps, "warmup": warmup, "ms_per_step": sec * 1e3, "dtype": mode,
           "dominant_kernel": table[0] if table else None, "other_kernels": table[1:6],
           "parity": slot_parity(kind, dev) if parity else None}
    torch.cuda.empty_cache()
    return res


def mode_run(precision, dev, cfg, frames_h, levels_h, B, gidx, mode_parity):
    """One in-tolerance precision mode measured exactly like `value`: the same 30-frame clip host to host through
    restore.restore_clip_single4x_host, warm, 3 timed steps; its dominant kernel's roofline from one profiled
    HBM-resident step; its parity on the bench tile."""
    import torch
    from elvis_amd import ops, restore
    model = restore.get_sinsr_model(dev, cfg=cfg, precision=precision)
    F = frames_h.shape[0]
    out_h = torch.empty_like(frames_h).pin_memory()
    batch = 6
    step = lambda: restore.restore_clip_single4x_host(model, frames_h, levels_h, B, gidx, out_h, batch=batch)
    sec = _timed_steps(step, 1, 3)
    prof = []
    ops.CONV_PROFILER = prof
    el = _timed_steps(step, 0, 1)
    ops.CONV_PROFILER = None
    table = _kernel_table(prof, [], el, "f16", conv_mfma_factor=ops.x3_mfma_factor)
    tot_fl = sum(w for _, w, _, _ in prof)
    tot_s = sum(e0.elapsed_time(e1) for _, _, e0, e1 in prof) * 1e-3
    res = {"mode": precision, "value": F / sec, "unit": "frames/s", "steps": 3, "warmup": 1, "ms_per_step": sec * 1e3,
           "frames_per_invocation": batch, "timed_region": "host to host, as `value`",
           "roofline": dict(table[0], note="peak = dense f16 MFMA; a compensated kernel executes `mfma_per_algorithmic` f16 MFMAs per "
                                           "algorithmic one: `achieved` / `frac` count algorithmic FLOPs, `executed_mfma_*` the matrix work") if table else None,
           "second_kernel": table[1] if len(table) > 1 else None,
           "all_conv_kernels": {"achieved": tot_fl / tot_s / 1e12, "time_share_of_step": tot_s / el},
           "parity": mode_parity}
    torch.cuda.empty_cache()
    return res, out_h


def staged_run(dev, cfg, frames_h, levels_h, B, first, nframes=6):
    """The drop-in DEFAULT schedule (restore_frames_sinsr(schedule="staged"), the reference's coarse-to-fine loop
    elvis.py:2570-2598 with 4x stages) on 1080p / block 8 / levels 0-3, through the P2 surface itself: a list of numpy
    frames in, a list of numpy frames out.  From level 3 that is a 4x call on the /8 image, then a 4x call on the /2
    image (2160x3840 out, area-halved): ~4.3x the network work of `single4x`."""
    import numpy as np
    from elvis_amd import restore
    frames = [np.ascontiguousarray(f) for f in frames_h[:nframes].numpy()]
    maps = levels_h[:nframes].numpy()
    run = lambda: restore.restore_frames_sinsr(frames, maps, B, dev, cfg=cfg, first_frame_index=first, schedule="staged")
    run()
    t0 = time.perf_counter()
    out = run()
    sec = time.perf_counter() - t0
    t0 = time.perf_counter()
    restore.restore_frames_sinsr(frames, maps, B, dev, cfg=cfg, first_frame_index=first, schedule="single4x")
    sec1 = time.perf_counter() - t0
    return {"schedule": "staged (drop-in default): SR 4x from /8, then SR 4x from /2 area-halved", "frames": nframes,
            "value": nframes / sec, "unit": "frames/s", "surface": "restore_frames_sinsr: numpy frame list in / out (unpinned host memory)",
            "single4x_same_surface": nframes / sec1, "max_level": int(maps.max()),
            "level0_blocks_bit_identical": bool(all(np.array_equal(o[np.repeat(np.repeat(m == 0, B, 0), B, 1)],
                                                                    f[np.repeat(np.repeat(m == 0, B, 0), B, 1)])
                                                     for o, f, m in zip(out, frames, maps)))}


def staged_parity(dev):
    """Staged schedule vs glue_ref.upscale_adaptive driven by the ORACLE network (full-width config) on a 320x320 frame,
    block 8, levels 0-3: u8 LSB of the f16 default and of the in-tolerance modes."""
    import numpy as np
    import torch
    from elvis_amd import restore
    from elvis_amd.weights import SinSRConfig, frame_noise, make_sinsr_weights
    from oracle import glue_ref, sinsr_ref as R
    cfg = SinSRConfig()
    sd = make_sinsr_weights(cfg, 0)
    rng = np.random.default_rng(7)
    S = 320   # /8 -> 40x40 (reflect-padded to 64 inside the UNet: the pad must stay below the size), x4 -> 160, x4 -> 640, /2 -> 320
    base = rng.random((S // 4 + 2, S // 4 + 2, 3))
    frame = np.clip(np.kron(base, np.ones((4, 4, 1)))[:S, :S] * 255 + rng.normal(0, 4, (S, S, 3)), 0, 255).astype(np.uint8)
    levels = rng.integers(0, 4, size=(1, S // 8, S // 8)).astype(np.int32)
    levels[0, 0, 0] = 3

    def up(img_bgr):
        lr = torch.from_numpy(np.ascontiguousarray(img_bgr[:, :, ::-1]))
        hp, wp = R.padded_latent_shape(cfg, lr.shape[0], lr.shape[1])
        o = R.sinsr_forward(sd, cfg, lr, frame_noise(cfg, restore.DEFAULT_SEED, 0, hp, wp))
        return np.ascontiguousarray(R.to_u8(o).numpy()[:, :, ::-1])

    ref = glue_ref.upscale_adaptive(frame, levels[0], 8, up, step=4).astype(int)
    out = {"tile": f"{S}x{S} frame, block 8, levels 0-3, full-width config; oracle = glue_ref.upscale_adaptive + the CPU network"}
    for prec in (None, "dec_f16", "x3"):
        got = restore.restore_frames_sinsr([frame], levels, 8, dev, cfg=cfg, schedule="staged", precision=prec)[0].astype(int)
        d = np.abs(got - ref)
        out[prec or "f16"] = {"max_abs_u8": int(d.max()), "frac_gt_1lsb": float((d > 1).mean())}
    return out


# ----------------------------------------------------------------------------- one rank
def main():
    args = parse()
    launch_ranks_if_needed(args)
    if args.launcher_selftest:
        return launcher_selftest(args)
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before the first GPU call
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl":
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {ndev} GPU(s) visible")
    local_dev = local_rank % ndev              # (gloo rehearsal: ranks may share a device)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    use_dist = world > 1 or (args.force_dist and "RANK" in os.environ)
    if use_dist:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        world = dist.get_world_size()          # n_gpus in the line = the ranks the backend saw

    from elvis_amd import ops, restore, synth
    from elvis_amd.weights import SinSRConfig

    F, H, W, B = args.frames, args.height, args.width, args.block
    if args.slot != "downsample":
        if rank == 0:
            s = slot_run(args.slot, dev, args.mode, frames=F, steps=max(1, args.steps), warmup=max(1, args.warmup),
                         parity=not args.no_cpu_baseline)
            print(json.dumps({"metric": f"restored 1080p frames/sec, ELVIS v2 {args.slot} path", "value": s["frames_per_s"],
                              "unit": "frames/s", "n_gpus": 1, "steps": s["steps"], "warmup": s["warmup"], "ms_per_step": s["ms_per_step"],
                              "higher_is_better": True, "dtype": args.mode, "data": "synthetic", "config": {"workload": s["workload"]},
                              "roofline": s["dominant_kernel"], "other_kernels": s["other_kernels"], "parity": s["parity"]}))
        return
    cfg = SinSRConfig()
    model = restore.get_sinsr_model(dev, cfg=cfg, fp32=(args.mode == "f32"), fuse_gn=args.fuse_gn)

    # ---- synthetic clip (host, untimed): a few distinct frames tiled to F to keep setup short
    # (clean clip and level maps on the host; the server-side degrade - per block INTER_AREA down by 2**level,
    # INTER_LINEAR back, elvis.py:2141-2169 - runs on the device: elvis_amd.degrade)
    from elvis_amd import degrade
    base_frames = min(F, 6)
    clean = synth.synth_clip(synth.CLIP_SEED, base_frames, H, W)
    levels = np.minimum(synth.synth_level_maps(synth.MAP_SEED, base_frames, H // B, W // B), 3).astype(np.int32)
    degraded = degrade.degrade_downsample_device(torch.from_numpy(clean).to(dev), torch.from_numpy(levels).to(dev), B).cpu().numpy()
    reps = (F + base_frames - 1) // base_frames
    frames_h = torch.from_numpy(np.concatenate([degraded] * reps)[:F]).pin_memory()
    levels_h = torch.from_numpy(np.concatenate([levels] * reps)[:F].astype(np.int32)).pin_memory()
    out_h = torch.empty((F, H, W, 3), dtype=torch.uint8).pin_memory()
    first = rank * F
    gidx = list(range(first, first + F))
    gathered = torch.empty((world * F, H, W, 3), dtype=torch.uint8, device=dev) if use_dist else None
    gathered_h = torch.empty((world * F, H, W, 3), dtype=torch.uint8).pin_memory() if use_dist and rank == 0 else None

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- primary: host -> host
    gather_dl = torch.cuda.Stream(dev) if use_dist else None   # rank 0's download of the gathered sequence: a copy stream of its
                                                               # own, so that it overlaps the next step's kernels (1.5 GB at N = 8)

    def step_host():
        _, shard_d = restore.restore_clip_single4x_host(model, frames_h, levels_h, B, gidx, out_h, batch=args.batch,
                                                        want_device=True)
        if use_dist:   # one all-gather of the restored clip; rank 0 brings the whole sequence to the host
            cur = torch.cuda.current_stream(dev)
            cur.wait_stream(gather_dl)                 # the previous step's download still reads `gathered`
            dist.all_gather_into_tensor(gathered, shard_d)
            if rank == 0:
                gather_dl.wait_stream(cur)
                with torch.cuda.stream(gather_dl):
                    gathered_h.copy_(gathered, non_blocking=True)

    for _ in range(args.warmup):
        step_host()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_host()
    fence()
    elapsed = time.perf_counter() - t0

    # ---- secondary: HBM-resident (round-1 definition), with per-launch HIP events for the roofline
    frames_d, levels_d = frames_h.to(dev), levels_h.to(dev)
    noise_d = model.make_noise(restore.DEFAULT_SEED, gidx, H // 4, W // 4)
    out_d = torch.empty_like(frames_d)
    active = (levels_h > 0).flatten(1).any(dim=1).tolist()

    def step_dev():
        restore.restore_clip_single4x_device(model, frames_d, levels_d, B, gidx, noise=noise_d, out=out_d,
                                             batch=args.batch, active=active)
        if use_dist:
            torch.cuda.current_stream(dev).wait_stream(gather_dl)
            dist.all_gather_into_tensor(gathered, out_d)

    step_dev()
    fence()
    same = bool(torch.equal(out_d.cpu(), out_h))      # both forms restore the same frames, bit for bit
    steps2 = max(1, min(args.steps, 5))
    prof = None if args.no_kernel_timing else []
    ops.CONV_PROFILER = prof
    t0 = time.perf_counter()
    for _ in range(steps2):
        step_dev()
    fence()
    elapsed2 = time.perf_counter() - t0
    ops.CONV_PROFILER = None
    gather_ok = None
    if use_dist and rank == 0:   # the gathered sequence on the host: rank 0's own shard must sit at its global offset, bit for bit
        gather_ok = bool(torch.equal(gathered_h[first:first + F], out_h))
    if use_dist:
        t = torch.tensor([elapsed, elapsed2], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed2 = float(t[0].item()), float(t[1].item())

    # ---- roofline of the dominant kernel from the live HIP events
    roof = None
    if prof:
        agg = _aggregate(prof)
        name, (fl, sec, cnt) = max(agg.items(), key=lambda kv: kv[1][1])
        tot_fl = sum(a[0] for a in agg.values())
        tot_s = sum(a[1] for a in agg.values())
        peak = PEAK_TFLOPS[args.mode]
        # HBM bytes per launch of this kernel: NOT measured in this run - read from the committed PMC passes of
        # the same workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, tools/pmc_traffic.py);
        # null when no file of this round names this kernel
        traffic = traffic_src = None
        tpath = os.path.join(ROOT, TRAFFIC_FILE)
        if os.path.exists(tpath) and args.batch == 15 and (H, W) == (1080, 1920):
            rec = json.load(open(tpath))["kernels"].get(name)
            if rec:
                traffic = rec["hbm_bytes_per_launch"]
                traffic_src = f"from file {TRAFFIC_FILE} (rocprofv3 PMC, separate passes of this workload; not measured in this run)"
        roof = {"bound": "mfma", "kernel": name, "achieved": fl / sec / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": fl / sec / 1e12 / peak, "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_source": traffic_src, "launches": cnt,
                "avg_launch_ms": sec / cnt * 1e3, "flop_per_launch_avg": fl / cnt,
                "all_conv_kernels": {"achieved": tot_fl / tot_s / 1e12, "time_share_of_step": tot_s / elapsed2,
                                     "tflop_per_frame": tot_fl / (steps2 * F) / 1e12}}

    if rank == 0:
        cpu_base = parity = slots = in_tol = staged = None
        extras = world == 1 and not args.no_extras and args.mode == "f16" and (H, W) == (1080, 1920)
        mode_par = {}
        if not args.no_cpu_baseline and world == 1:
            cpu_base, parity, mode_par = cpu_baseline_and_parity(args.mode, dev, args.fuse_gn,
                                                                 ("x3", "dec_f16") if extras else ())
        if extras:
            # the precision modes that meet the north star's tolerance, measured exactly like `value`
            in_tol = {"note": "x3 = fp32 tensors, every conv on the f16 matrix pipe with the operand rounding compensated (<= 1e-4 "
                              "max-abs: inside the 1e-3 bar everywhere); dec_f16 = encoder + Swin-UNet as x3, VQ lookup, decoder in f16 "
                              "(no code flips: the u8 frame is within 1 LSB of the CPU path's with the lookup on; its f32 max-abs is "
                              "the decoder's f16 arithmetic alone)"}
            for prec in ("x3", "dec_f16"):
                in_tol[prec], o = mode_run(prec, dev, cfg, frames_h, levels_h, B, gidx, mode_par.get(prec))
                in_tol[prec]["u8_max_abs_vs_f16_headline_output"] = int((o.to(torch.int16) - out_h.to(torch.int16)).abs().max())
            staged = staged_run(dev, cfg, frames_h, levels_h, B, first)
            staged["parity"] = staged_parity(dev)
        if not args.no_slots and world == 1 and not args.no_extras:
            slots = {k: slot_run(k, dev, args.mode, frames=F) for k in ("dct", "blur")}
        total_frames = world * F * args.steps
        line = {
            "metric": "restored 1080p frames/sec, SinSR 4x path (ELVIS v2 Downsample)",
            "value": total_frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.mode, "data": "synthetic",
            "config": {"workload": f"{F}-frame {W}x{H} synthetic clip per GPU, ELVIS v2 Downsample (SinSR 4x), "
                                   f"block {B}, levels 0-3, weights seed 0 (random init), quantize=True",
                       "timed_region": "host to host: pinned host frames+maps -> H2D, sampler noise (host cores), network, "
                                       "recompose" + (", one RCCL all-gather, rank 0 D2H of the whole sequence" if world > 1 else ", D2H")
                                       + " -> pinned host frames",
                       "frames_per_gpu": F,
                       "parallelism": (f"frame-sharded x{world}, 1 all-gather ({'RCCL' if args.backend == 'nccl' else 'gloo REHEARSAL on ' + str(ndev) + ' GPU(s) - not a scaling measurement'})"
                                       if world > 1 else "single GPU"),
                       "gn_fused_into_conv": bool(args.fuse_gn), "frames_per_invocation": args.batch},
            "hbm_resident": {"value": world * F * steps2 / elapsed2, "unit": "frames/s", "steps": steps2,
                             "ms_per_step": elapsed2 / steps2 * 1e3, "same_output_as_host_path": same,
                             "note": "inputs, noise and outputs resident in HBM (round-1 definition); never `value`"},
            "roofline": roof, "cpu_baseline": cpu_base, "parity": parity, "in_tolerance": in_tol, "staged_schedule": staged,
            "slots": slots,
            "dist": ({"backend": ("RCCL (torch 'nccl')" if args.backend == "nccl" else "gloo"), "world": world, "forced_at_world_1": bool(args.force_dist and world == 1),
                      "collectives_per_step": "one all_gather_into_tensor of the restored uint8 clip",
                      "gathered_sequence_holds_rank0_shard_bit_for_bit": gather_ok} if use_dist else None),
        }
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
