#!/usr/bin/env python3
"""Benchmark of the ELVIS v2 Downsample (SinSR 4x) client-side restore hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one clip per rank: `--frames` (default 30) degraded
1080p uint8 frames + block maps, RESIDENT IN HBM when the timed region starts ->
/4 area downscale -> SinSR 4x (VQ-f4 encode, Swin-UNet single step, VQ-f4 decode) -> block-map
recompose -> restored uint8 frames in HBM (+ one RCCL all-gather of the restored clip when N>1).
Weak scaling: every rank owns its own 30-frame clip (the chunk_for_devices split of a 30*N-frame
clip, elvis.py:255-280).  value = frames restored by all ranks / max-over-ranks step time.

The JSON line also carries
  roofline     - the dominant kernel (implicit-GEMM conv on MFMA): algorithmic FLOPs / HIP-event
                 time measured live in the timed region on the launch stream, vs the dense f16
                 MFMA peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s; fp32 MFMA 157.3 TFLOP/s)
  cpu_baseline - the CPU oracle (PyTorch fp32 restatement, oracle/sinsr_ref.py) timed on this
                 host's cores on ONE 256x256 output tile, scaled to 1080p frames/s
  parity       - max-abs / PSNR of the benchmarked mode against the CPU oracle on that tile
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=30, help="frames per rank per step")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--block", type=int, default=8)
    ap.add_argument("--mode", choices=["f16", "f32"], default="f16")
    ap.add_argument("--no-fuse-gn", dest="fuse_gn", action="store_false",
                    help="run GroupNorm-apply+SiLU as separate elementwise passes instead of the conv prologue")
    ap.add_argument("--batch", type=int, default=15, help="frames per network invocation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


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


def cpu_baseline_and_parity(model_mode, dev, fuse_gn):
    """CPU baseline: the oracle timed on a 768x768 output region of the full-width config (scaled to
    1080p frames/s by pixel count).  Parity: GPU vs oracle on a 256x256 output tile (64x64 LR)."""
    import dataclasses
    from elvis_amd.sinsr import SinSRModel
    from elvis_amd.weights import SinSRConfig, frame_noise, make_sinsr_weights
    from oracle import sinsr_ref as R

    cfg = SinSRConfig()
    sd = make_sinsr_weights(cfg, 0)
    from elvis_amd.synth import synth_clip
    lr = torch.from_numpy(synth_clip(20260501, 1, 64, 64)[0])
    noise = frame_noise(cfg, 42, 0, 64, 64)
    cores = host_cores()
    torch.set_num_threads(cores)
    R.ae_encode(sd, cfg, torch.zeros(1, 3, 32, 32))  # warm the thread pool / oneDNN primitives
    # timed sample: a 768x768 output region (192x192 LR, 9 tiles' worth, ~10-20 s of CPU work)
    S = 192
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
    ref = R.sinsr_forward(sd, cfg, lr, noise)
    dt = torch.float16 if model_mode == "f16" else torch.float32

    def compare(c, ref_img):
        model = SinSRModel(c, sd, dev, dt, fuse_gn=fuse_gn)
        u8, f32 = model.forward(lr[None].to(dev), noise.to(dev), want_f32=True)
        u8b, _ = model.forward(lr[None].to(dev), noise.to(dev), want_f32=True)
        ref_u8 = R.to_u8(ref_img).numpy().astype(np.float32)
        got = u8[0].cpu().numpy().astype(np.float32)
        mse = float(np.mean((got - ref_u8) ** 2))
        out = {"max_abs_f32": float((f32[0].cpu() - ref_img).abs().max()),
               "psnr_db_vs_oracle_u8": float("inf") if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse)),
               "max_abs_u8": float(np.abs(got - ref_u8).max()),
               "bit_reproducible": bool(torch.equal(u8, u8b))}
        del model
        return out

    # continuous path (no VQ lookup) is the stable parity figure; with the (discontinuous) lookup
    # in the loop a code flip anywhere shows up as a local O(0.1) difference
    cfg_c = dataclasses.replace(cfg, quantize=False)
    ref_c = R.sinsr_forward(sd, cfg_c, lr, noise)
    parity = {"tile": "256x256 output, full-width config, weights seed 0, noise seed 42",
              "continuous_path": compare(cfg_c, ref_c), "with_vq_lookup": compare(cfg, ref)}
    torch.cuda.empty_cache()
    return base, parity


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from elvis_amd import ops, restore, synth
    from elvis_amd.weights import SinSRConfig

    F, H, W, B = args.frames, args.height, args.width, args.block
    cfg = SinSRConfig()
    model = restore.get_sinsr_model(dev, cfg=cfg, fp32=(args.mode == "f32"), fuse_gn=args.fuse_gn)

    # ---- synthetic clip (host, untimed): a few distinct frames tiled to F to keep setup short
    base_frames = min(F, 6)
    clean, degraded, levels = synth.make_downsample_case(base_frames, H, W, B, max_level=3)
    reps = (F + base_frames - 1) // base_frames
    degraded = np.concatenate([degraded] * reps)[:F]
    levels = np.concatenate([levels] * reps)[:F]
    frames_d = torch.from_numpy(degraded).to(dev)
    levels_d = torch.from_numpy(levels.astype(np.int32)).to(dev)
    first = rank * F
    gidx = list(range(first, first + F))
    noise_d = model.make_noise(restore.DEFAULT_SEED, gidx, H // 4, W // 4)
    out_d = torch.empty_like(frames_d)
    gathered = torch.empty((world * F, H, W, 3), dtype=torch.uint8, device=dev) if world > 1 else None

    def step():
        restore.restore_clip_single4x_device(model, frames_d, levels_d, B, gidx, noise=noise_d, out=out_d,
                                             batch=args.batch)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out_d)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    prof = None
    if not args.no_kernel_timing:
        prof = []
        ops.CONV_PROFILER = prof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.CONV_PROFILER = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel from the live HIP events
    roof = None
    if prof:
        agg = {}
        for name, flops, e0, e1 in prof:
            a = agg.setdefault(name, [0.0, 0.0, 0])
            a[0] += flops
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        name, (fl, sec, cnt) = dom
        tot_fl = sum(a[0] for a in agg.values())
        tot_s = sum(a[1] for a in agg.values())
        peak = PEAK_TFLOPS[args.mode]
        # HBM bytes per launch of this kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs of this same workload, tools/pmc_traffic.py); null if none is on file
        traffic = traffic_src = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_k_hbm_traffic.json")
        if os.path.exists(tpath) and args.batch == 15 and (args.height, args.width) == (1080, 1920):
            rec = json.load(open(tpath))["kernels"].get(name)
            if rec:
                traffic, traffic_src = rec["hbm_bytes_per_launch"], "profiles/r01_k_hbm_traffic.json (PMC, separate passes)"
        roof = {"bound": "mfma", "kernel": name, "achieved": fl / sec / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": fl / sec / 1e12 / peak, "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_source": traffic_src, "launches": cnt,
                "avg_launch_ms": sec / cnt * 1e3, "flop_per_launch_avg": fl / cnt,
                "all_conv_kernels": {"achieved": tot_fl / tot_s / 1e12, "time_share_of_step": tot_s / elapsed,
                                     "tflop_per_frame": tot_fl / (args.steps * F) / 1e12}}

    if rank == 0:
        cpu_base = parity = None
        if not args.no_cpu_baseline and world == 1:
            cpu_base, parity = cpu_baseline_and_parity(args.mode, dev, args.fuse_gn)
        total_frames = world * F * args.steps
        line = {
            "metric": "restored 1080p frames/sec, SinSR 4x path (ELVIS v2 Downsample)",
            "value": total_frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.mode, "data": "synthetic",
            "config": {"workload": f"{F}-frame {W}x{H} synthetic clip per GPU, ELVIS v2 Downsample (SinSR 4x), "
                                   f"block {B}, levels 0-3, weights seed 0 (random init), quantize=True",
                       "frames_per_gpu": F, "parallelism": f"frame-sharded x{world}, 1 all-gather" if world > 1 else "single GPU",
                       "gn_fused_into_conv": bool(args.fuse_gn), "frames_per_invocation": args.batch},
            "roofline": roof, "cpu_baseline": cpu_base, "parity": parity,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
