#!/usr/bin/env python3
"""bench.py -- stabilised frames/s on synthetic NV12 clips (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k|1080p] [--mode auto|warp]

One "step" = one output frame of the hot path on this rank's clip.  Inputs (a ring of distinct
synthetic NV12 frames) are resident in HBM before the timed region.  N > 1: launched by
torch.distributed.run, one rank per GPU, one independent clip per rank (weak scaling, no
data-path collective), timing = max over ranks between barriers.

Rank 0 prints ONE JSON line with the driver's contract plus
  "roofline":     the fused undistort-remap kernel's algorithmic bytes / measured launch time
                  (HIP events on the launch stream) against the 8 TB/s HBM peak, and
  "cpu_baseline": the CPU oracle (a port of the reference's cvtColor -> createMap -> remap path)
                  timed on a bounded sample on this host's cores.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="4k", choices=["4k", "1080p"])
    ap.add_argument("--mode", default="auto", choices=["auto", "warp", "pipeline"])
    ap.add_argument("--ring", type=int, default=64, help="distinct input frames / output buffers")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic", default=None, help="measured HBM bytes per launch from PMC passes (profiles/)")
    return ap.parse_args()


def synth_ring(torch, dev, w, h, n, seed):
    """n distinct NV12 frames generated on the device: smooth luma texture + blocks, smooth chroma."""
    g = torch.Generator(device=dev).manual_seed(1234 + seed)
    frames = []
    yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    base = torch.rand((h // 8 + 2, w // 8 + 2), device=dev, generator=g)
    tex = torch.nn.functional.interpolate(base[None, None], size=(h, w), mode="bilinear", align_corners=False)[0, 0]
    for i in range(n):
        f = torch.empty((h * 3 // 2, w), dtype=torch.uint8, device=dev)
        shift = torch.roll(tex, shifts=(3 * i, 5 * i), dims=(0, 1))
        noise = torch.rand((h, w), device=dev, generator=g) * 0.08
        f[:h] = ((0.15 + 0.7 * shift + noise).clamp(0, 1) * 255).to(torch.uint8)
        cu = 128 + 50 * torch.sin(xx[::2, ::2] / w * 6.0 + i * 0.1)
        cv = 128 + 50 * torch.cos(yy[::2, ::2] / h * 5.0 + i * 0.07)
        uv = torch.stack([cu, cv], dim=-1).to(torch.uint8)
        f[h:] = uv.reshape(h // 2, w)
        frames.append(f)
    return frames


def cpu_baseline(w, h, K, Ko, cw, ch, budget_s=12.0):
    """The reference's CPU path (cvtColor -> createMap -> remap, FrameSourceWarp.cpp:401,272-314)
    as restated in oracle/vstab_oracle.c, timed on this host on a bounded sample of frames."""
    import oracle
    import synth
    frame = synth.nv12(0, w, h)
    p = oracle.map_params(K, Ko, np.eye(3))
    # host share of a 1-GPU box is 16 CPUs: never oversubscribe beyond the affinity mask
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))
    oracle.lib().vo_set_num_threads(threads)
    oracle.warp_nv12(frame, p, cw, ch)  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.warp_nv12(frame, p, cw, ch)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    return {"value": round(n / el, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} frames of {w}x{h} NV12 -> {cw}x{ch} BGR, undistort-remap only "
                      f"(cvtColor+createMap+remap, identity rotation), OpenMP over rows, {el:.1f} s"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    vs = importlib.import_module("video-annotator_amd")  # raises if libvstab.so is missing: no fallback
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    w, h = (3840, 2160) if args.workload == "4k" else (1920, 1080)
    preset = vs.GOPRO_H4B_WIDE169_MEASURED
    K = vs.get_preset_camera(preset, w, h)
    Ko, (cw, ch) = vs.get_output_camera(K, w, h, 1.0, False, 1.0)

    have_pipeline = hasattr(vs, "Stabilizer")
    mode = args.mode
    if mode == "auto":
        mode = "pipeline" if have_pipeline else "warp"

    ring = synth_ring(torch, dev, w, h, args.ring, seed=rank) if mode == "warp" else None
    outs = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(args.ring)]
    stream = torch.cuda.current_stream()

    # a fixed per-frame rotation schedule (small smooth shake) so every launch has a different map
    def rot(i):
        a = 0.02 * np.sin(0.11 * i), 0.015 * np.cos(0.07 * i), 0.01 * np.sin(0.05 * i + 1)
        th = np.linalg.norm(a)
        k = np.array(a) / th if th > 0 else np.zeros(3)
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx

    kernel_events = []
    if mode == "warp":
        params = [vs.map_params(K, Ko, rot(i)) for i in range(args.warmup + args.steps)]

        def step(i, timed):
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            vs.warp_nv12_bgr(ring[i % args.ring], params[i], cw, ch, out=outs[i % args.ring])
            if timed:
                e1.record(stream)
                kernel_events.append((e0, e1))
        workload = f"{args.workload} NV12 {w}x{h} -> BGR {cw}x{ch}, fused undistort-remap (createMap+cvtColor+remap), per-frame rotation, tracking/smoothing NOT included"
    else:
        stab = vs.Stabilizer.synthetic_clip(dev, w, h, preset=preset, seed=1234 + rank, ring=args.ring,
                                            frames=args.warmup + args.steps + 64)

        def step(i, timed):
            stab.pull_into(outs[i % args.ring], timing=kernel_events if timed else None)
        workload = f"{args.workload} NV12 {w}x{h} -> BGR {cw}x{ch}, full pipeline: NV12 ingest, corner detect, pyramidal LK, rotation estimate, SG smoothing (r=30), fused undistort-remap"

    for i in range(args.warmup):
        step(i, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        kms = [a.elapsed_time(b) for a, b in kernel_events]
        avg_ms = float(np.mean(kms)) if kms else None
        alg_bytes = w * h * 1.5 + cw * ch * 3  # NV12 read once + BGR8 written once
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms else None
        traffic = None
        if args.traffic:
            traffic = float(args.traffic)
        else:
            tf = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
            if os.path.exists(tf):
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
        line = {
            "metric": "stabilized frames/sec at 4K NV12, 1/2/4/8 GPU; remap % HBM roofline",
            "value": round(world * args.steps / el, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8 pixels / f32 map / f64 rotations", "data": "synthetic",
            "config": {"workload": workload, "mode": mode, "clips": world, "ring_frames": args.ring,
                       "preset": "GOPRO_H4B_WIDE169_MEASURED", "parallelism": f"clip-per-gpu x{world}"},
            "roofline": {"bound": "hbm", "kernel": "k_warp_tiled", "achieved": round(achieved, 1) if achieved else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "traffic": traffic, "algorithmic_bytes_per_launch": int(alg_bytes),
                         "avg_launch_us": round(avg_ms * 1e3, 2) if avg_ms else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, h, K, Ko, cw, ch)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
