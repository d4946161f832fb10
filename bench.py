#!/usr/bin/env python3
"""bench.py -- stabilised frames/s on synthetic NV12 clips (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k|1080p] [--mode auto|warp]

One "step" = one pass of the hot path over one batch of synthetic input: --batch consecutive frames of
this rank's clip (default 256 in pipeline mode: four passes over the 64-frame input ring), each emitted
as one output frame; `value` is frames/s, computed from the whole timed region; `step_ms` carries the
median / min / max of the steps' GPU-side durations (events at the step boundaries).  Inputs (a ring of distinct synthetic NV12 frames) are resident in HBM before the timed
region, and an untimed pre-roll
(--preroll, default 1024 frames) ahead of --warmup fills the look-ahead queue and brings clocks,
prefetch depth and the speculative detector to steady state.  N > 1: one rank per GPU, one
independent clip per rank (weak scaling, no data-path collective), timing
= max over ranks between barriers.  `bench.py --gpus N` without a torch.distributed environment
starts the N ranks itself (torch.distributed.run as a child process, before any GPU call) and
relays rank 0's line; under torch.distributed.run WORLD_SIZE must equal --gpus.

Rank 0 prints ONE JSON line with the driver's contract plus
  "roofline":     the fused undistort-remap kernel's algorithmic bytes / its average duration in the timed region --
                  the kernel's own start / end stamps (hipExtLaunchKernelGGL events on its launch stream, every 8th
                  launch; what rocprofv3's kernel trace reports) -- against the 8 TB/s HBM peak, and
  "cpu_baseline": the CPU oracle (a port of the reference's cvtColor -> createMap -> remap path)
                  timed on a bounded sample on this host's cores (all the box's share; "cpu_baseline_1_thread": one), and
  "copy_ingest":  the same pipeline with every frame COPIED into the library's ring (vstab_frame.hold = 0, what a decoder
                  that recycles its surface gives), measured after the timed region in the same windowing (steps x batch), and
  "host":         what the rank costs the host: CPU seconds (user + system, all threads) per 1000 frames of the timed region, and
  "cpu_baseline_full_pipeline": the oracle's whole consume_frame / pull_frame loop (detector, LK, smoothing, warp), and
  "parity_check": one frame emitted after the timed region compared bit for bit with the oracle's warp of the same input
                  frame under the rotation the pipeline reports for it (the run fails if they differ).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
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
    ap.add_argument("--steps", type=int, default=40, help="timed steps; one step = one batch of --batch frames")
    ap.add_argument("--warmup", type=int, default=4, help="untimed steps ahead of the timed region")
    ap.add_argument("--batch", type=int, default=None,
                    help="frames per step.  Default: 256 in pipeline mode (four passes over the input ring; 20 steps are ~190 ms at 4K, so that one "
                         "host stall does not move the line), 64 (one pass over the ring) with --mode warp")
    ap.add_argument("--preroll", type=int, default=1024, help="untimed frames ahead of --warmup (pipeline mode): steady state before the driver's window")
    ap.add_argument("--fixed-preroll", action="store_true", help="exactly --preroll untimed frames (default: at least that many, then until the rate is steady)")
    ap.add_argument("--workload", default="4k", choices=["4k", "1080p", "4k-p010"],
                    help="4k-p010 = BASELINE config 5: P010 frames, 10-bit pixel path with fp16 blend, a read-out rotation per frame")
    ap.add_argument("--mode", default="auto", choices=["auto", "warp", "pipeline"])
    ap.add_argument("--ring", type=int, default=64, help="distinct input frames / output buffers")
    ap.add_argument("--no-tracking", action="store_true",
                    help="BASELINE config 1: undistort only (identity rotations) through the pipeline object")
    ap.add_argument("--out-format", default="bgr", choices=["bgr", "nv12", "p010", "nv12-planar", "p010-planar"],
                    help="bgr = what FrameSourceWarp emits (the BASELINE metric); nv12 = encoder hand-off through BGR (the BGR frame converted in the "
                         "kernel); p010 = the same for --workload 4k-p010; nv12-planar / p010-planar = SURVEY.md 8(f) row 2 as written: the planes "
                         "remapped as they are, no colour round trip (vstab_pull_frame_nv12_planar / vstab_pull_frame_p010_planar)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the driver's multi-GPU runs); gloo only for rehearsing ranks on one GPU")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--rehearse-launcher", action="store_true",
                    help="functional rehearsal of the multi-rank plumbing WITHOUT any GPU work (gloo on the CPU): start the ranks, rendezvous, pin "
                         "every rank to its own CPUs, barriers, the MAX all-reduce, the all-gather of the per-clip records, the concat list. "
                         "The line carries value null: nothing is measured.  (A GPU box admits six processes on its card, so N = 8 can only be "
                         "rehearsed like this; the real 1 -> 8 curve is the driver's.)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: initialise torch.distributed and run the collectives even with one rank (exercises the RCCL path on one GPU)")
    ap.add_argument("--ingest", default="inplace", choices=["inplace", "copy"],
                    help="inplace = upstream holds its frames (vstab_frame.hold), planes are read where they are, no pack kernel; "
                         "copy = hold 0: every frame goes through vstab_pack_nv12 into the library's ring")
    ap.add_argument("--map-precision", default="opencl", choices=["ieee", "opencl"],
                    help="opencl (default, as in vstab_config_default) = the arithmetic of the reference's own kernel, createMap.cl as ROCm's OpenCL "
                         "compiler builds it for gfx950 (bit-identical to that code object, which is also the parity checker of the run); "
                         "ieee = createMap.cl with every operation IEEE-rounded (CPU-reproducible; checked against the CPU oracle)")
    ap.add_argument("--skip-ieee-pass", action="store_true",
                    help="leave out the short extra pass that reports the IEEE-map rate beside the default run's (key `ieee_map`)")
    ap.add_argument("--pull", choices=("batch", "single"), default="single",
                    help="pipeline mode, BGR output: one vstab_pull_frame call per frame from this Python loop (default), or one "
                         "vstab_pull_frames call per step (the consumer's loop in C, as the reference's DisplayImage.cpp runs it); "
                         "measured equal within the run-to-run noise at 4K (DESIGN.md section 6)")
    ap.add_argument("--skip-copy-pass", action="store_true",
                    help="leave out the extra copy-ingest pass behind the timed region (used when the command runs under rocprofv3, so that the "
                         "kernel population of the trace is the timed pipeline's)")
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


def shaky_ring(torch, dev, w, h, K, n, seed):
    """n NV12 frames of a static spherical scene seen through the equidistant fisheye camera K while
    the camera orientation follows a smooth PERIODIC shake (period n, so the ring can be cycled
    without a jump): about 0.2 degrees per frame per axis.  Rendered on
    the GPU with torch (harness only).  Returns (frames, rotations)."""
    g = torch.Generator(device=dev).manual_seed(1234 + seed)
    th, tw = 4096, 8192
    tex = torch.zeros((th, tw), device=dev)
    for cell, amp in ((16, 0.5), (48, 0.3), (160, 0.2)):
        base = torch.rand((th // cell + 2, tw // cell + 2), device=dev, generator=g)
        tex += amp * torch.nn.functional.interpolate(base[None, None], size=(th, tw), mode="bilinear", align_corners=False)[0, 0]
    tex = 40 + 120 * tex
    rs = np.random.default_rng(1234 + seed)
    for _ in range(6000):  # bright / dark rectangles: trackable corners everywhere on the sphere
        rw, rh = rs.integers(20, 90), rs.integers(20, 90)
        x, y = rs.integers(0, tw - rw), rs.integers(0, th - rh)
        tex[y:y + rh, x:x + rw] = float(rs.choice([225.0, 30.0, 200.0]))
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    px, py = (xs - float(K[0, 2])) / float(K[0, 0]), (ys - float(K[1, 2])) / float(K[1, 1])
    ang = torch.sqrt(px * px + py * py)
    sc = torch.where(ang > 1e-9, torch.sin(ang) / ang.clamp_min(1e-9), torch.ones_like(ang))
    rays = torch.stack([px * sc, py * sc, torch.cos(ang)], dim=-1)  # d_cam
    cu = (128 + 50 * torch.sin(xs[::2, ::2] / w * 6.0)).to(torch.uint8)
    cv = (128 + 50 * torch.cos(ys[::2, ::2] / h * 5.0)).to(torch.uint8)
    uv = torch.stack([cu, cv], dim=-1).reshape(h // 2, w)
    amp = rs.uniform(0.008, 0.02, (3, 3))
    ph = rs.uniform(0, 2 * np.pi, (3, 3))
    frames, rots = [], []
    for k in range(n):
        a = [sum(amp[ax, j] * np.sin(2 * np.pi * (j + 1) * k / n + ph[ax, j]) for j in range(3)) for ax in range(3)]
        t = float(np.linalg.norm(a))
        kk = np.array(a) / t if t > 0 else np.zeros(3)
        Kx = np.array([[0, -kk[2], kk[1]], [kk[2], 0, -kk[0]], [-kk[1], kk[0], 0]])
        R = np.cos(t) * np.eye(3) + (1 - np.cos(t)) * np.outer(kk, kk) + np.sin(t) * Kx
        d = rays @ torch.tensor(R, dtype=torch.float32, device=dev)  # rows: R^T d_cam = d_world
        lon, lat = torch.atan2(d[..., 0], d[..., 2]), torch.asin(d[..., 1].clamp(-1, 1))
        u, v = (lon / (2 * np.pi) + 0.5) * (tw - 1), (lat / np.pi + 0.5) * (th - 1)
        u0, v0 = u.long().clamp(0, tw - 2), v.long().clamp(0, th - 2)
        fu, fv = u - u0, v - v0
        val = (tex[v0, u0] * (1 - fu) + tex[v0, u0 + 1] * fu) * (1 - fv) + (tex[v0 + 1, u0] * (1 - fu) + tex[v0 + 1, u0 + 1] * fu) * fv
        f = torch.empty((h * 3 // 2, w), dtype=torch.uint8, device=dev)
        f[:h] = val.round().clamp(0, 255).to(torch.uint8)
        f[h:] = uv
        frames.append(f)
        rots.append(R)
    return frames, rots


GPUS_PER_NODE = 8  # an MI355X node carries eight GPUs: one GPU's fair share of the host is an eighth of its CPUs


def box_cpu_share():
    """Threads the CPU legs may use: one GPU's share of the host -- its CPUs / 8 ("the same box's host cores";
    a 256-CPU node gives 32) -- and never more than this process is allowed to run on."""
    return max(1, min(len(os.sched_getaffinity(0)), max(1, (os.cpu_count() or 1) // GPUS_PER_NODE)))


def share_txt():
    return (f"{box_cpu_share()} threads = min(CPUs this process may use: {len(os.sched_getaffinity(0))}, host CPUs / {GPUS_PER_NODE} GPUs per node: "
            f"{(os.cpu_count() or 1) // GPUS_PER_NODE})")


def cpu_baseline(w, h, K, Ko, cw, ch, budget_s=10.0, threads=None):
    """The reference's CPU path (cvtColor -> createMap -> remap, FrameSourceWarp.cpp:401,272-314)
    as restated in oracle/vstab_oracle.c, timed on this host on a bounded sample of frames."""
    import oracle
    import synth
    frame = synth.nv12(0, w, h)
    p = oracle.map_params(K, Ko, np.eye(3))
    threads = box_cpu_share() if threads is None else threads
    oracle.lib().vo_set_num_threads(threads)
    oracle.warp_nv12(frame, p, cw, ch)  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.warp_nv12(frame, p, cw, ch)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 2000:
            break
    return {"value": round(n / el, 3), "unit": "frames/s", "cores": threads, "kind": "port", "share": share_txt() if threads > 1 else "1 thread",
            "sample": f"{n} frames of {w}x{h} NV12 -> {cw}x{ch} BGR, undistort-remap only "
                      f"(cvtColor+createMap+remap, identity rotation), OpenMP over rows, {el:.1f} s"}


def cpu_baseline_planar(frame, w, h, K, Ko, cw, ch, depth, blend=0, budget_s=10.0, threads=None):
    """The plane-wise warp's CPU restatement (createMap -> remap of the luma plane -> remap of the chroma plane: oracle/vstab_oracle.c
    vo_warp_planar_mapped), timed like cpu_baseline; frame = the NV12 frame (depth 8) or the stacked P010 planes as uint16 (depth 10)."""
    import oracle
    p = oracle.map_params(K, Ko, np.eye(3))
    threads = box_cpu_share() if threads is None else threads
    oracle.lib().vo_set_num_threads(threads)
    run = (lambda: oracle.warp_nv12_planar(frame, p, cw, ch, 0)) if depth == 8 else (lambda: oracle.warp_p010_planar(frame[:h], frame[h:], p, cw, ch, 0, None, blend))
    run()  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        run()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 2000:
            break
    what = "NV12 -> NV12 planes" if depth == 8 else "P010 -> P010 planes, " + ("binary16" if blend else "exact") + " blend"
    return {"value": round(n / el, 3), "unit": "frames/s", "cores": threads, "kind": "port", "share": share_txt() if threads > 1 else "1 thread",
            "sample": f"{n} frames of {w}x{h} {what} {cw}x{ch}, plane-wise undistort-remap only (createMap + remap of the luma plane + remap of the "
                      f"chroma plane, identity rotation), OpenMP over rows, {el:.1f} s"}


def cpu_baseline_p010(frame16, w, h, K, Ko, cw, ch, budget_s=10.0):
    """The oracle's definition of the config-5 warp (10-bit conversion, per-row map, fp16 blend) on the host cores."""
    import oracle
    p = oracle.map_params(K, Ko, np.eye(3))
    rb = oracle.map_params(K, Ko, oracle.rodrigues((0.002, 0.001, -0.001)))[8:]
    y, uv = frame16[:h], frame16[h:]
    threads = box_cpu_share()
    oracle.lib().vo_set_num_threads(threads)
    oracle.warp_p010(y, uv, p, cw, ch, rb, 0, 1)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        oracle.warp_p010(y, uv, p, cw, ch, rb, 0, 1)
        n += 1
    el = time.perf_counter() - t0
    return {"value": round(n / el, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} frames of the oracle's 10-bit warp chain (conversion, per-row map planes, fp16 blend) in {el:.1f} s"}


def cpu_baseline_full(frames, K, Ko, cw, ch, w, h, budget_s=12.0):
    """The whole reference loop on the host, restated (oracle): key-frame corner detection, pyramidal LK, SG smoothing,
    cvtColor -> createMap -> remap -- consume_frame / pull_frame of FrameSourceWarp.cpp:397-476 on frames copied back
    from the GPU ring.  The rotation estimate itself (a5, microseconds of fp64 work) is left out: identity rotations."""
    import oracle
    threads = box_cpu_share()
    oracle.lib().vo_set_num_threads(threads)

    def track(prev, cur, corners):
        if len(corners) == 0:
            return corners, corners
        nxt, st = oracle.pyr_lk(prev, cur, corners)
        return corners[st > 0], nxt[st > 0]

    r = 2  # look-ahead only delays the first output; it does not change the per-frame work
    sm = oracle.WarpStateMachine(frames, r, lambda g: oracle.good_features(np.ascontiguousarray(g)), track,
                                 lambda pp, cp: (np.eye(3), 100), lambda f, R: oracle.warp_nv12(f, oracle.map_params(K, Ko, R), cw, ch))
    n, t0 = 0, time.perf_counter()
    while sm.pull_frame() is not None:
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    return {"value": round(n / el, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} frames of {w}x{h}: corner detection on key frames, pyramids + LK (single-threaded per call), SG filter, "
                      f"cvtColor+createMap+remap (OpenMP), {el:.1f} s"}


def side_pass(torch, pull, args, what):
    """A pipeline variant behind the timed region, in the driver's windowing: 1024 frames of pre-roll, --warmup steps, then --steps steps
    of --batch frames between synchronisations, with an event at every step boundary."""
    for i in range(1024 + args.warmup * args.batch):
        assert pull(i)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True)]
    tc = time.perf_counter()
    evs[0].record()
    for k in range(args.steps):
        for i in range(args.batch):
            assert pull(k * args.batch + i)
        evs.append(torch.cuda.Event(enable_timing=True))
        evs[-1].record()
    torch.cuda.synchronize()
    el = time.perf_counter() - tc
    ms = [a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:])]
    return {"value": round(args.steps * args.batch / el, 1), "unit": "frames/s", "steps": args.steps, "frames_per_step": args.batch,
            "step_ms": {"median": round(float(np.median(ms)), 4), "min": round(min(ms), 4), "max": round(max(ms), 4)},
            "what": what + f"; {args.steps} steps of {args.batch} frames after a pre-roll of 1024 frames + {args.warmup} steps, behind the timed region"}


def launch_ranks(args):
    """`bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child process (nothing in this process has
    touched the GPU), relay rank 0's JSON line and the exit code.  Model: concat.sh:248 (xargs -P N)."""
    # (no device query here: the launcher stays provably free of the GPU runtime; a rank without a GPU exits 2 by itself)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def _numa_cpus(torch, dev_index):
    """CPUs of the NUMA node the GPU hangs off (sysfs), or None."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        path = f"/sys/bus/pci/devices/{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}.0/numa_node"
        node = int(open(path).read())
        if node < 0:
            return None, None
        ids = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            ids.update(range(int(a), int(b or a) + 1))
        return node, sorted(ids)
    except Exception:
        return None, None


PIN_CPUS = 8  # one handle runs four busy threads (caller, estimate worker, corner-selection helper, runtime): a CCX-sized set keeps them on one L3


def pin_rank(local, rank, world, share_gpu, torch):
    """Keep this rank's threads (the library's helper threads inherit the mask) on a few CPUs of its GPU's NUMA node: the node's
    CPUs are dealt in equal consecutive shares to the ranks whose GPU hangs off that node (so the ranks of a node never share a
    core), and a rank takes the first PIN_CPUS of its share.  Without sysfs information: an equal share of the allowed CPUs.
    Unpinned, the 1080p rate on a two-socket host was bimodal (31 k / 36 k / 39 k frames/s: profiles/r04_cpu_pinning_1080p.txt).
    Returns (description, previous mask)."""
    allowed = sorted(os.sched_getaffinity(0))
    if os.environ.get("VSTAB_BENCH_PIN", "1") == "0":
        return f"{len(allowed)} CPUs (unpinned)", allowed
    node, cpus = _numa_cpus(torch, local)
    if cpus:
        cpus = [c for c in cpus if c in set(allowed)]
    if cpus:
        if share_gpu:     # rehearsal: every rank on one card -> the node's CPUs in `world` disjoint shares
            peers, k = world, rank
        elif world == 1:  # one GPU visible: a node of an 8-GPU box serves 8 / (number of NUMA nodes) GPUs; take the first share
            nodes = max(1, len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]))
            peers, k = max(1, GPUS_PER_NODE // nodes), 0
        else:             # the local ranks whose GPU hangs off the same node share it
            same = [r for r in range(world) if _numa_cpus(torch, r)[0] == node]
            peers, k = max(1, len(same)), same.index(local) if local in same else 0
        share = max(1, len(cpus) // peers)
        cpus, how = cpus[k * share:(k + 1) * share][:PIN_CPUS] or cpus[:PIN_CPUS], f"numa node {node}, share {k + 1} of {peers}"
    else:
        share = max(1, len(allowed) // max(1, world))
        cpus, how = (allowed[rank * share:(rank + 1) * share] or allowed)[:PIN_CPUS], "equal share of the allowed CPUs"
    try:
        os.sched_setaffinity(0, cpus)
        return f"CPUs {cpus[0]}-{cpus[-1]} ({len(cpus)}; {how})", allowed
    except OSError:
        return f"{len(allowed)} CPUs (unpinned)", allowed


def rehearse_launcher(args, rank, local, world, torch, dist):
    """--rehearse-launcher: everything around the GPU work of an N-rank run, on the CPU over gloo (see parse())."""
    class _NoCuda:  # pin_rank asks torch.cuda for the GPU's PCI address; there is none here -> equal shares of the allowed CPUs
        class cuda:
            @staticmethod
            def get_device_properties(i):
                raise RuntimeError("no GPU in a launcher rehearsal")
    pinned, _ = pin_rank(local, rank, world, True, _NoCuda)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("gloo")
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rank % 3))  # stands in for the timed region; ranks finish at different times
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    shard = importlib.import_module("video-annotator_amd.shard")
    mine = sorted(os.sched_getaffinity(0))
    n_frames = args.steps * (args.batch or 256)
    rec = dict(rank=rank, clip=rank, frames=n_frames, elapsed_ns=int(float(t.item()) * 1e9), crc=shard.crc_of(np.full(16, rank, np.uint8)), cpu_first=mine[0],
               cpu_last=mine[-1], cpu_count=len(mine))
    records = shard.gather_records([rec], device=torch.device("cpu"))
    if rank == 0:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        listing = shard.concat_list([f"clip_{r['clip']:02d}_stabilised.mp4" for r in records])
        with open(os.path.join(ROOT, "gpurun_out", f"concat_list_{world}gpu_rehearsal.txt"), "w") as fh:
            fh.write(listing)
        print(json.dumps({"metric": "stabilized frames/sec at 4K NV12, 1/2/4/8 GPU; remap % HBM roofline", "value": None, "unit": "frames/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak (unmeasured: launcher rehearsal)",
                          "rehearsal": "launcher only: ranks, rendezvous, CPU pinning, barriers, MAX all-reduce, record all-gather and concat list ran over gloo on "
                                       "the CPU; no GPU work was done and nothing was measured",
                          "config": {"workload": "none (launcher rehearsal)", "clips": len(records), "parallelism": f"clip-per-gpu x{world}"},
                          "records": len(records), "frames_per_clip": sorted({r["frames"] for r in records}), "concat_list_lines": listing.count("\n"),
                          "rank_cpus": pinned, "rank_cpus_all": [f"{r['cpu_first']}-{r['cpu_last']} ({r['cpu_count']})" for r in records],
                          "collectives": "gloo"}), flush=True)
    dist.destroy_process_group()
    return 0


def main():
    args = parse()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report {world} rank(s) as {args.gpus} GPUs", file=sys.stderr)
        return 2
    if args.rehearse_launcher:
        return rehearse_launcher(args, rank, local, world, torch, dist)
    if not args.share_gpu and local >= torch.cuda.device_count():
        print(f"bench.py: --gpus {args.gpus}: rank {rank} has no GPU (LOCAL_RANK {local}, {torch.cuda.device_count()} visible; "
              "use --share-gpu --dist-backend gloo to rehearse ranks on one GPU)", file=sys.stderr)
        return 2
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pinned, cpu_mask = pin_rank(local, rank, world, args.share_gpu, torch)
    vs = importlib.import_module("video-annotator_amd")  # raises if libvstab.so is missing: no fallback
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # where collective payloads live
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when the communicator is created; stdout is reserved for the one JSON
        # line, so the banner is sent to stderr (fd-level redirect: it comes from C code)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group("gloo")
            dist.barrier()  # creates the communicator now
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    w, h = (1920, 1080) if args.workload == "1080p" else (3840, 2160)
    p010 = args.workload == "4k-p010"
    if p010 and (args.mode == "warp" or args.out_format in ("nv12", "nv12-planar") or args.no_tracking):
        print("bench.py: --workload 4k-p010 runs the full pipeline with 16-bit BGR or P010 output", file=sys.stderr)
        return 2
    if args.out_format in ("p010", "p010-planar") and not p010:
        print("bench.py: --out-format p010 / p010-planar belongs to --workload 4k-p010", file=sys.stderr)
        return 2
    planar = args.out_format in ("nv12-planar", "p010-planar")   # the plane-wise warp: no colour round trip
    p010_out = p010 and args.out_format in ("p010", "p010-planar")
    preset = vs.GOPRO_H4B_WIDE169_MEASURED
    K = vs.get_preset_camera(preset, w, h)
    Ko, (cw, ch) = vs.get_output_camera(K, w, h, 1.0, False, 1.0)

    have_pipeline = hasattr(vs, "Stabilizer")
    mode = args.mode
    if mode == "auto":
        mode = "pipeline" if have_pipeline else "warp"

    if args.batch is None:
        args.batch = 256 if mode == "pipeline" else 64
    ring = synth_ring(torch, dev, w, h, args.ring, seed=rank) if mode == "warp" else None
    nv12_out = args.out_format in ("nv12", "nv12-planar")
    out_fmt = vs.OUT_NV12_PLANAR if planar else vs.OUT_NV12 if nv12_out else vs.OUT_BGR8
    if nv12_out:
        outs = [vs.nv12_out_planes(cw, ch, dev) for _ in range(args.ring)]
    elif p010_out:
        outs = [(torch.empty((ch, cw), dtype=torch.int16, device=dev), torch.empty(((ch + 1) // 2, 2 * ((cw + 1) // 2)), dtype=torch.int16, device=dev))
                for _ in range(args.ring)]
    elif p010:
        outs = [torch.empty((ch, cw, 3), dtype=torch.int16, device=dev) for _ in range(args.ring)]
    else:
        outs = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(args.ring)]
    out_name = "NV12 (plane-wise warp, no colour round trip)" if planar else "NV12 (the BGR frame converted)" if nv12_out else "BGR"
    if os.environ.get("VSTAB_BENCH_OWN_STREAM"):  # development: the caller works on a stream of its own instead of the default stream (the default stream stays in the process: a fifth stream)
        torch.cuda.set_stream(torch.cuda.Stream())
    stream = torch.cuda.current_stream()

    # a fixed per-frame rotation schedule (small smooth shake) so every launch has a different map
    def rot(i):
        a = 0.02 * np.sin(0.11 * i), 0.015 * np.cos(0.07 * i), 0.01 * np.sin(0.05 * i + 1)
        th = np.linalg.norm(a)
        k = np.array(a) / th if th > 0 else np.zeros(3)
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx

    opencl = args.map_precision == "opencl"
    map_mode = vs.MAP_CREATEMAP_CL_OPENCL if opencl else vs.MAP_CREATEMAP_CL
    kernel_events = []
    preroll = 0
    n_warm, n_timed = args.warmup * args.batch, args.steps * args.batch  # frames
    if mode == "warp":
        params = [vs.map_params(K, Ko, rot(i)) for i in range((args.warmup + args.steps) * args.batch)]

        def step(i, timed):
            if timed and i % 8 == 0:   # the kernel's own start / end stamps (hipExtLaunchKernelGGL), every 8th launch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                vs.time_next_launch(e0, e1)
                kernel_events.append((e0, e1))
            vs.warp_nv12(ring[i % args.ring], params[i], cw, ch, map_mode, out_fmt, out=outs[i % args.ring])
        workload = f"{args.workload} NV12 {w}x{h} -> {out_name} {cw}x{ch}, fused undistort-remap (createMap+cvtColor+remap), per-frame rotation, tracking/smoothing NOT included"
    else:
        clip, _ = shaky_ring(torch, dev, w, h, K, args.ring, seed=rank)
        preroll = max(0, args.preroll)
        extra, readouts = {}, None
        if p010:
            # the 8-bit ring widened to P010: its bytes on top (so the tracker, which sees the narrowed luma, does what it
            # does in the 8-bit run), two more bits of detail, junk in the six unused bits; plus the camera's rotation
            # during each frame's read-out (a tenth of the per-frame shake schedule)
            g = torch.Generator(device=dev).manual_seed(77 + rank)
            clip = [((f.to(torch.int32) << 8) | torch.randint(0, 256, f.shape, generator=g, device=dev, dtype=torch.int32)).to(torch.int16) for f in clip]
            readouts = [rot(i) @ rot(i + 1).T for i in range(len(clip))]
            readouts = [np.eye(3) + 0.1 * (R - np.eye(3)) for R in readouts]
            readouts = [np.linalg.svd(R)[0] @ np.linalg.svd(R)[2] for R in readouts]   # back onto SO(3)
            extra = dict(bit_depth=10, readouts=readouts, pixel_depth=10, blend=vs.BLEND_FP16)
        extra["map_precision"] = vs.MAP_PRECISION_OPENCL if opencl else vs.MAP_PRECISION_IEEE
        ring_hold = 0 if args.ingest == "copy" else None
        stab = vs.Stabilizer(clip, total=preroll + 32 * 256 + (args.warmup + args.steps) * args.batch + 1000, preset=preset, smooth_radius=30, seed=1234 + rank,
                             tracking=0 if args.no_tracking else 1, ring_hold=ring_hold, **extra)
        assert stab.out_size == (cw, ch)

        def step(i, timed):
            if timed and i == n_warm:
                stab.profile()               # fold + discard the warm-up stages
                stab._prof0 = stab.profile()
            assert pull(i)
        def make_pull(st):
            if p010:
                if planar:
                    return lambda i: st.pull_p010_planar_into(*outs[i % args.ring])
                return (lambda i: st.pull_p010_into(*outs[i % args.ring])) if p010_out else (lambda i: st.pull_bgr16_into(outs[i % args.ring]))
            return (lambda i: st.pull_nv12_into(*outs[i % args.ring], planar=planar)) if nv12_out else (lambda i: st.pull_into(outs[i % args.ring]))
        pull = make_pull(stab)
        stab.enable_profiling(1)  # timed region: HIP events around the warp launches only
        ingest_txt = ("frames used in place (upstream holds them: vstab_frame.hold), no pack kernel" if args.ingest == "inplace"
                      else "every frame copied into the library's ring (vstab_frame.hold = 0: vstab_pack_nv12)")
        prec_txt = (", map in the arithmetic of the reference's own createMap kernel (createMap.cl built for gfx950 by ROCm's OpenCL compiler; bit-identical)"
                    if opencl else ", map with every operation IEEE-rounded (--map-precision ieee: CPU-reproducible, not the reference's GPU arithmetic)")
        workload = (f"{args.workload} NV12 {w}x{h} -> {out_name} {cw}x{ch}, full pipeline: {ingest_txt}; corner detect, pyramidal LK, rotation estimate, "
                    f"SG smoothing (r=30), fused undistort-remap{prec_txt}")
        if args.no_tracking:
            workload = (f"{args.workload} NV12 {w}x{h} -> {out_name} {cw}x{ch}, undistort only (tracking off, identity rotations): {ingest_txt}; "
                        f"fused undistort-remap from the map written once{prec_txt}")
        if p010:
            out10 = "P010 planes (plane-wise warp, no colour round trip)" if planar else "P010 planes" if p010_out else "BGR 16-bit (10 significant)"
            workload = (f"4k P010 {w}x{h} -> {out10} {cw}x{ch}, BASELINE config 5: full pipeline on the narrowed luma (corner detect, "
                        f"pyramidal LK, rotation estimate, SG r=30), 10-bit undistort-remap with fp16 blend and a rotation per output row{prec_txt}")

    # pipeline mode, BGR frames: a step is ONE library call that pulls the step's batch of frames into the output ring
    # (vstab_pull_frames: the consumer's frame loop on the C side of the boundary, where the reference has it)
    batched = mode == "pipeline" and args.pull == "batch" and not nv12_out and not p010
    # every step boundary gets an event on the caller's stream (recorded, never waited for inside the region): the spread of the
    # steps' GPU-side durations travels in the line, so that one slow step is told from a slower run
    step_events = []

    def mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        step_events.append(e)
    if batched:
        def run_steps(first_frame, n_steps, timed):
            for k in range(n_steps):
                i0 = first_frame + k * args.batch
                if timed and k == 0:
                    stab.profile()               # fold + discard the warm-up stages
                    stab._prof0 = stab.profile()
                assert stab.pull_frames_into(outs, i0, args.batch) == args.batch
                if timed:
                    mark()
    else:
        def run_steps(first_frame, n_steps, timed):
            for k in range(n_steps):
                for i in range(first_frame + k * args.batch, first_frame + (k + 1) * args.batch):
                    step(i, timed)
                if timed:
                    mark()
    for i in range(preroll):  # pipeline mode: untimed, ahead of the warm-up the driver asks for
        assert pull(i)
    if mode == "pipeline" and preroll and not args.fixed_preroll:
        # still untimed: on a box that has just been started the first seconds run slower (clocks, page-in of the runtime) -- a 4K
        # run that opened its 51 ms window right behind 1024 frames read 22 k where the next run read 26.4 k.  Keep pulling in
        # windows of 256 frames until two consecutive windows agree within 3 % (at most 32 windows); the frames are counted in `preroll`.
        prev, more = None, 0
        for _ in range(32):
            torch.cuda.synchronize()
            tw = time.perf_counter()
            for i in range(256):
                assert pull(preroll + more + i)
            torch.cuda.synchronize()
            rate = 256 / (time.perf_counter() - tw)
            more += 256
            if prev is not None and abs(rate - prev) <= 0.03 * prev:
                break
            prev = rate
        preroll += more
    import gc
    gc.collect()
    gc.disable()  # no collector pause of the Python frame loop inside the ~50 ms window (nothing is skipped: the loop allocates a few small objects per frame)
    run_steps(0, args.warmup, False)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    mark()
    run_steps(n_warm, args.steps, True)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    cpu1 = time.process_time()
    el_own = el  # this rank's own wall time (el becomes the maximum over the ranks below)
    gc.enable()
    step_ms = [a.elapsed_time(b) for a, b in zip(step_events[:-1], step_events[1:])]
    # what a rank costs the HOST: CPU seconds (user + system, every thread of this process: the caller's frame loop, the handle's
    # estimate worker and corner-selection helper, the runtime's threads) per 1000 frames of the timed region
    cpu_s = cpu1 - cpu0
    try:
        n_threads = int([ln.split()[1] for ln in open("/proc/self/status") if ln.startswith("Threads:")][0])
    except Exception:
        n_threads = None
    host_cost = {"cpu_seconds_per_1000_frames": round(cpu_s / max(1, n_timed) * 1000, 4), "cpu_seconds": round(cpu_s, 4), "wall_seconds": round(el_own, 4),
                 "cpus_busy": round(cpu_s / el_own, 2), "threads": n_threads, "pinned_cpus": len(os.sched_getaffinity(0)),
                 "what": "user + system time of every thread of this rank's process over the timed region (time.process_time); cpus_busy = that / "
                         "wall time: the CPUs one rank keeps busy (its frame loop + the handle's estimate worker and corner-selection helper, which "
                         "spin before they sleep) -- the figure that says whether N ranks fit one host; seconds per 1000 frames = cpus_busy / rate"}
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # correctness tie-in (outside the timed region): the next frame the pipeline emits, against the oracle's warp of the
    # same input frame under the rotation the pipeline reports for it (FrameSourceWarp.cpp:471-475); bit for bit
    parity, parity_against = None, None
    if mode == "pipeline" and rank == 0:
        import oracle
        # the checker's OpenMP threads are created on first use and keep the CPU mask of that moment for good: give them this GPU's
        # whole share of the host (the CPU baseline legs below run on the same pool), then go back to the handle's few CPUs
        pinned_mask = sorted(os.sched_getaffinity(0))
        try:
            os.sched_setaffinity(0, cpu_mask)
            oracle.lib().vo_set_num_threads(box_cpu_share())
            oracle.cvt_nv12_bgr(np.zeros((96, 64), np.uint8))  # a first parallel region: the pool exists from here on
            os.sched_setaffinity(0, pinned_mask)
        except OSError:
            pass
        n_emit = preroll + n_warm + n_timed  # index of the frame pulled now; frame 0 of the clip is never emitted
        assert pull(n_emit)
        torch.cuda.synchronize()
        src = clip[(n_emit + 1) % len(clip)].cpu().numpy()
        pr = oracle.map_params(K, Ko, stab.warp_rotation(n_emit))
        got = outs[n_emit % args.ring]
        if p010 and planar:
            rb = oracle.map_params(K, Ko, readouts[(n_emit + 1) % len(clip)] @ stab.warp_rotation(n_emit))[8:]
            s16 = src.view(np.uint16)
            ey, euv = (oracle.warp_p010_planar_ref_gfx950(s16[:h], s16[h:], pr, cw, ch, rb, 1) if opencl
                       else oracle.warp_p010_planar(s16[:h], s16[h:], pr, cw, ch, 0, rb, 1))
            same = np.array_equal(got[0].cpu().numpy().view(np.uint16), ey) and np.array_equal(got[1].cpu().numpy().view(np.uint16), euv)
        elif planar:
            ey, euv = oracle.warp_nv12_planar_ref_gfx950(src, pr, cw, ch) if opencl else oracle.warp_nv12_planar(src, pr, cw, ch, 0)
            same = np.array_equal(got[0].cpu().numpy(), ey) and np.array_equal(got[1].cpu().numpy(), euv)
        elif p010:
            rb = oracle.map_params(K, Ko, readouts[(n_emit + 1) % len(clip)] @ stab.warp_rotation(n_emit))[8:]
            s16 = src.view(np.uint16)
            # opencl: the reference kernel's map row by row (one launch per output row with that row's matrix), oracle conversion and remap
            exp16 = oracle.warp_p010_ref_gfx950(s16[:h], s16[h:], pr, cw, ch, rb, 1) if opencl else oracle.warp_p010(s16[:h], s16[h:], pr, cw, ch, rb, 0, 1)
            if p010_out:
                ey, euv = oracle.cvt_bgr10_p010(exp16)
                same = np.array_equal(got[0].cpu().numpy().view(np.uint16), ey) and np.array_equal(got[1].cpu().numpy().view(np.uint16), euv)
            else:
                same = np.array_equal(got.cpu().numpy().view(np.uint16), exp16)
        elif nv12_out and not opencl:
            exp_y, exp_uv = oracle.warp_nv12_ex(src, pr, cw, ch, 0, 1)
            same = np.array_equal(got[0].cpu().numpy().reshape(-1), exp_y.reshape(-1)) and np.array_equal(got[1].cpu().numpy().reshape(-1), exp_uv.reshape(-1))
        elif opencl:
            # the checker of this mode is the reference's own kernel (oracle/_ref/createMap.gfx950.co) run on this GPU, then
            # the oracle's cvtColor and cv::remap (and its BGR -> NV12 for the encoder hand-off format)
            exp = oracle.warp_nv12_ref_gfx950(src, pr, cw, ch)
            if nv12_out:
                exp_y, exp_uv = oracle.cvt_bgr_nv12(exp)
                same = np.array_equal(got[0].cpu().numpy().reshape(-1), exp_y.reshape(-1)) and np.array_equal(got[1].cpu().numpy().reshape(-1), exp_uv.reshape(-1))
            else:
                same = np.array_equal(got.cpu().numpy(), exp)
        else:
            same = np.array_equal(got.cpu().numpy(), oracle.warp_nv12(src, pr, cw, ch))
        parity = "ok" if same else "MISMATCH"
        parity_against = ("the reference's createMap kernel (oracle/_ref/createMap.gfx950.co, run on this GPU) -> oracle "
                          + ("plane-wise remap (vo_remap_plane)" if planar else "cvtColor / remap") if opencl else "the CPU oracle's IEEE chain")

    # end-of-run record exchange: the run's only collective (RCCL all-gather over xGMI when N > 1),
    # then rank 0 writes the stitch list (concat.sh / join.ts format)
    shard = importlib.import_module("video-annotator_amd.shard")
    last = outs[(n_warm + n_timed - 1) % args.ring]
    if nv12_out or p010_out:
        last = last[0]
    mine = sorted(os.sched_getaffinity(0))
    rec = dict(rank=rank, clip=rank, frames=n_timed, elapsed_ns=int(el_own * 1e9), crc=shard.crc_of(last[:64].cpu().numpy()), cpu_first=mine[0], cpu_last=mine[-1],
               cpu_count=len(mine), cpu_us=int(cpu_s * 1e6))
    records = shard.gather_records([rec], device=cdev)
    if rank == 0 and world > 1:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"concat_list_{world}gpu.txt"), "w") as fh:
            fh.write(shard.concat_list([f"clip_{r['clip']:02d}_stabilised.mp4" for r in records]))

    if rank == 0:
        kms = [a.elapsed_time(b) for a, b in kernel_events]
        avg_ms = float(np.mean(kms)) if kms else None
        stages = None
        if mode == "pipeline":
            p1, p0 = stab.profile(), stab._prof0
            d = {k: p1[k] - p0[k] for k in p1}
            avg_ms = d["gpu_warp_ms"] / max(1, d["warp_timed"])  # every 8th warp launch: the kernel's own start / end stamps
            stages = {k: round(v / max(1, d["frames_emitted"]) * 1e3, 2) for k, v in d.items() if k.endswith("_ms")}
            stages = {k.replace("_ms", "_us_per_frame"): v for k, v in stages.items()}
            stages["key_frames"] = int(d["key_frames"])
            timed_stages = {k: v for k, v in stages.items() if k.startswith("host_") or k == "key_frames"}
            timed_stages["gpu_warp_us_per_timed_launch"] = round(avg_ms * 1e3, 2)
            timed_stages["warp_launches_timed"] = int(d["warp_timed"])
            # per-stage table from a short extra pass with every stage timed (outside the timed region)
            stab.enable_profiling(2)
            q0 = stab.profile()
            for i in range(120):
                assert pull(i)
            q1 = stab.profile()
            dq = {k: q1[k] - q0[k] for k in q1}
            stages = {k.replace("_ms", "_us_per_frame"): round(v / max(1, dq["frames_emitted"]) * 1e3, 2) for k, v in dq.items() if k.endswith("_ms")}
            stages["key_frames_per_120"] = int(dq["key_frames"])
        # the same kernel with nothing beside it (no tracker / pyramid / detection kernels on other streams): back-to-back
        # launches on this stream, every launch with its own start / end stamps -- the figure that isolates kernel quality
        # from co-scheduling
        alone_us, copy_ingest, ieee_map = None, None, None
        if mode == "pipeline":
            pa = vs.map_params(K, Ko, rot(7))
            run_alone = lambda i: vs.warp_nv12(clip[i % len(clip)], pa, cw, ch, map_mode, out_fmt, out=outs[i % args.ring])  # cycling inputs and outputs: nothing stays in the caches
            if p010:
                pb = vs.map_params(K, Ko, rot(8))[8:]
                run_alone = lambda i: vs.warp_p010(clip[i % len(clip)][:h], clip[i % len(clip)][h:], pa, cw, ch, pb, vs.MAP_CREATEMAP_CL, vs.BLEND_FP16,
                                                   out=outs[i % args.ring])
                if p010_out:
                    fn10 = vs.warp_p010_planar if planar else vs.warp_p010_planes
                    run_alone = lambda i: fn10(clip[i % len(clip)][:h], clip[i % len(clip)][h:], pa, cw, ch, pb, vs.MAP_CREATEMAP_CL, vs.BLEND_FP16,
                                               out_y=outs[i % args.ring][0], out_uv=outs[i % args.ring][1])
            for i in range(10):
                run_alone(i)
            torch.cuda.synchronize()
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
            for i, (a, b) in enumerate(pairs):
                vs.time_next_launch(a, b)
                run_alone(10 + i)
            torch.cuda.synchronize()
            alone_us = float(np.mean([a.elapsed_time(b) for a, b in pairs])) * 1e3
            if args.ingest == "inplace" and not p010 and world == 1 and not args.skip_copy_pass:
                # what a decoder that recycles its surface gives: hold = 0, every frame through vstab_pack_nv12 into the ring
                # (untimed region; a short run of its own: pre-roll 200 frames, 10 batches)
                stab.close()  # two handles alive = eight streams on the runtime's four hardware queues, which slows both
                cstab = vs.Stabilizer(clip, total=1024 + (args.warmup + args.steps) * args.batch + 100, preset=preset, smooth_radius=30, seed=1234 + rank,
                                      tracking=0 if args.no_tracking else 1, ring_hold=0, **extra)
                copy_ingest = side_pass(torch, make_pull(cstab), args,
                                        "the same pipeline with vstab_frame.hold = 0: every frame copied into the library's ring by vstab_pack_nv12 "
                                        f"(+{int(w * h * 3)} B of traffic and one kernel per frame)")
                cstab.close()
            if opencl and not p010 and world == 1 and not args.skip_ieee_pass and not args.skip_copy_pass:
                # the same pipeline with the CPU-reproducible map (every operation IEEE-rounded; vstab_config.map_precision = IEEE): the
                # rate beside the default's, from a short run of its own in the untimed region
                stab.close()
                istab = vs.Stabilizer(clip, total=1024 + (args.warmup + args.steps) * args.batch + 100, preset=preset, smooth_radius=30, seed=1234 + rank,
                                      tracking=0 if args.no_tracking else 1, ring_hold=ring_hold, **dict(extra, map_precision=vs.MAP_PRECISION_IEEE))
                ieee_map = side_pass(torch, make_pull(istab), args,
                                     "the same pipeline with vstab_config.map_precision = IEEE (createMap.cl with every operation IEEE-rounded, reproducible by a "
                                     "CPU; not the reference's GPU arithmetic)")
                istab.close()
        alg_bytes = w * h * 1.5 + cw * ch * 3  # NV12 read once + BGR8 written once
        if nv12_out:
            alg_bytes = w * h * 1.5 + cw * ch + 2 * ((cw + 1) // 2) * ((ch + 1) // 2)
        kernel_name = "k_warp_planar<8, CREATEMAP_CL_OPENCL, DEPTH 8> (plane-wise: luma + chroma planes remapped as they are)" if planar else "k_warp_fused"
        if p010:
            alg_bytes = w * h * 3 + cw * ch * 6  # P010 read once + 16-bit BGR written once
            kernel_name = "k_warp_fused<8, RS_CREATEMAP_CL, BGR16, DEPTH 10, FP16> (the LDS-tiled kernel with a 10:10:10 LDS pixel)"
            if p010_out:
                alg_bytes = w * h * 3 + cw * ch * 2 + 4 * ((cw + 1) // 2) * ((ch + 1) // 2)  # P010 read once + P010 planes written once
                kernel_name = ("k_warp_planar<8, RS_CREATEMAP_CL, DEPTH 10, FP16> (plane-wise)" if planar
                               else "k_warp_fused<8, RS_CREATEMAP_CL, P010, DEPTH 10, FP16> (P010 planes written by the warp kernel)")
        base_bytes = alg_bytes  # of the kernel that evaluates the map (what "alone" runs)
        cached = mode == "pipeline" and args.no_tracking
        if cached:
            # tracking off: the map never changes, so the pipeline warps from the quantised map it wrote once -- the CACHED
            # instantiation, whose launch also reads 8 B of map per output pixel (vstab_quantised_map)
            kernel_name = "k_warp_fused<CACHED> (reads the quantised map)"
            alg_bytes += ((cw + 3) // 4 * 4) * ch * 8
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms else None
        # HBM traffic needs PMC passes under rocprofv3 (tools/prof_bench.sh); what this line can carry is the COMMITTED figure
        # of the same command (profiles/), named as such -- "traffic" itself is only set when the caller passes a measurement
        traffic = float(args.traffic) if args.traffic else None
        committed = None
        tf = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        prof = json.load(open(tf)) if os.path.exists(tf) else None
        if prof and not nv12_out and not cached and not p010 and prof.get("map_precision", "ieee") == args.map_precision:
            committed = {"traffic_bytes_per_launch": prof.get("hbm_bytes_per_launch"), "rocprof_avg_launch_us": prof.get("rocprof_avg_launch_us"),
                         "valu_busy": prof.get("valu_busy"), "source": f"profiles/traffic_{args.workload}.json ({prof.get('round', 'r02')}, --map-precision {prof.get('map_precision', 'ieee')}: separate rocprofv3 "
                                                                         "--kernel-trace / --pmc passes of this command on another box)"}
        line = {
            "metric": "stabilized frames/sec at 4K NV12, 1/2/4/8 GPU; remap % HBM roofline",
            "value": round(world * n_timed / el, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 5),
            "step_ms": {"median": round(float(np.median(step_ms)), 4), "min": round(min(step_ms), 4), "max": round(max(step_ms), 4),
                        "all": [round(v, 3) for v in step_ms] if len(step_ms) <= 64 else None,
                        "what": "GPU-side duration of each timed step: events on the caller's stream at the step boundaries.  The last steps of a run are shorter: "
                                "with frames used in place nothing holds the host and the tracker's streams back, they finish first, and the warps still "
                                "queued then have the GPU to themselves (value counts all of it, from the whole region)"} if step_ms else None,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": ("u10 pixels (fp16 blend)" if p010 else "u8 pixels") + " / f32 map / f64 rotations", "data": "synthetic",
            "config": {"workload": workload, "mode": mode, "clips": len(records), "ring_frames": args.ring, "frames_per_step": args.batch,
                       "frame_loop": ("vstab_pull_frames: one call per step" if batched else "vstab_pull_frame per frame from Python") if mode == "pipeline" else None,
                       "preset": "GOPRO_H4B_WIDE169_MEASURED", "parallelism": f"clip-per-gpu x{world}"},
            "rank_cpus_all": [f"{r.get('cpu_first', 0)}-{r.get('cpu_last', 0)} ({r.get('cpu_count', 0)})" for r in records] if world > 1 else None,
            "preroll": preroll, "parity_check": parity, "parity_check_against": parity_against if parity else None, "rank_cpus": pinned,
            "collectives": args.dist_backend if use_dist else None,
            "host": dict(host_cost, all_ranks_cpu_seconds_per_1000_frames=[round(r.get("cpu_us", 0) / 1e6 / max(1, r["frames"]) * 1000, 4) for r in records]
                         if world > 1 else None,
                         all_ranks_cpus_busy=[round(r.get("cpu_us", 0) * 1e3 / max(1, r["elapsed_ns"]), 2) for r in records] if world > 1 else None),
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1) if achieved else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "traffic": traffic, "algorithmic_bytes_per_launch": int(alg_bytes),
                         # kernel time in the timed region: the kernel's own start / end stamps (hipExtLaunchKernelGGL events on the
                         # launch stream), averaged over the launches timed -- the quantity rocprofv3 --kernel-trace reports
                         "avg_launch_us": round(avg_ms * 1e3, 2) if avg_ms else None,
                         "timing": "kernel start/end stamps (hipExtLaunchKernelGGL) of every 8th launch in the timed region",
                         "committed_profile": committed,
                         "alone": None if alone_us is None else {"avg_launch_us": round(alone_us, 2),
                                                                   "kernel": kernel_name.split(" (")[0],
                                                                   "achieved": round(base_bytes / alone_us / 1e3, 1),
                                                                   "frac": round(base_bytes / alone_us / 1e3 / HBM_PEAK_GBS, 4)}},
        }
        if copy_ingest:
            line["copy_ingest"] = copy_ingest
        if ieee_map:
            line["ieee_map"] = ieee_map
        if stages:
            line["stages"] = stages  # every GPU stage timed (extra pass outside the timed region; chained LK launches off)
            line["stages_timed_region"] = timed_stages  # host waits + warp launches as they were in the timed region
        try:
            os.sched_setaffinity(0, cpu_mask)  # the CPU legs run on this GPU's share of the host, not on the few CPUs the handle's threads were kept on
        except OSError:
            pass
        if world == 1 and not args.no_cpu_baseline and planar:  # the plane-wise outputs are priced against the plane-wise restatement
            f0 = clip[0].cpu().numpy()
            line["cpu_baseline"] = cpu_baseline_planar(f0.view(np.uint16) if p010 else f0, w, h, K, Ko, cw, ch, 10 if p010 else 8, blend=1 if p010 else 0)
            line["cpu_baseline_1_thread"] = cpu_baseline_planar(f0.view(np.uint16) if p010 else f0, w, h, K, Ko, cw, ch, 10 if p010 else 8, blend=1 if p010 else 0,
                                                                budget_s=6.0, threads=1)
        elif world == 1 and not args.no_cpu_baseline and p010:
            line["cpu_baseline"] = cpu_baseline_p010(clip[0].cpu().numpy().view(np.uint16), w, h, K, Ko, cw, ch)
        elif world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, h, K, Ko, cw, ch)
            line["cpu_baseline_1_thread"] = cpu_baseline(w, h, K, Ko, cw, ch, budget_s=6.0, threads=1)
            if mode == "pipeline" and not args.no_tracking:
                host_ring = [f.cpu().numpy() for f in clip]
                host_frames = [host_ring[i % len(host_ring)] for i in range(400)]  # bounded by the time budget inside
                line["cpu_baseline_full_pipeline"] = cpu_baseline_full(host_frames, K, Ko, cw, ch, w, h)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()
    if parity == "MISMATCH":
        print("bench.py: the emitted frame differs from the oracle", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
