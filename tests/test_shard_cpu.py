"""CPU tests of the multi-GPU path: clip -> rank assignment and the end-of-run record exchange
over a world_size-2 gloo group (the same code runs over RCCL on GPUs)."""
import importlib
import os
import socket

import numpy as np


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, q):
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    shard = importlib.import_module("video-annotator_amd.shard")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.assign_clips(n_clips, world)[rank]
    recs = [dict(rank=rank, clip=c, frames=100 + c, elapsed_ns=1000 * (c + 1), crc=shard.crc_of(np.full(16, c, np.uint8)))
            for c in mine]
    allrec = shard.gather_records(recs)
    q.put((rank, allrec))
    dist.barrier()
    dist.destroy_process_group()


def test_assign_clips_round_robin():
    shard = importlib.import_module("video-annotator_amd.shard")
    assert shard.assign_clips(8, 8) == [[i] for i in range(8)]
    assert shard.assign_clips(5, 2) == [[0, 2, 4], [1, 3]]
    assert shard.assign_clips(1, 4) == [[0], [], [], []]
    flat = sorted(c for r in shard.assign_clips(11, 3) for c in r)
    assert flat == list(range(11))


def test_concat_list_format():
    shard = importlib.import_module("video-annotator_amd.shard")
    assert shard.concat_list(["a.mp4", "dir/b c.mp4"]) == "file 'a.mp4'\nfile 'dir/b c.mp4'\n"   # join.ts:51-53
    assert shard.concat_list(["it's.mp4"]) == "file 'it'\\''s.mp4'\n"


def test_gather_records_world2_gloo():
    import torch.multiprocessing as mp
    shard = importlib.import_module("video-annotator_amd.shard")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_clips = 5   # uneven: rank 0 gets 3 clips, rank 1 gets 2
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]                                  # every rank holds the same clip-ordered list
    assert [r["clip"] for r in res[0]] == list(range(n_clips))
    assert [r["rank"] for r in res[0]] == [0, 1, 0, 1, 0]
    assert res[0][3]["frames"] == 103 and res[0][3]["crc"] == shard.crc_of(np.full(16, 3, np.uint8))


def test_gather_records_single_process():
    shard = importlib.import_module("video-annotator_amd.shard")
    recs = [dict(rank=0, clip=1, frames=2, elapsed_ns=3, crc=4), dict(rank=0, clip=0, frames=5, elapsed_ns=6, crc=7)]
    assert [r["clip"] for r in shard.gather_records(recs)] == [0, 1]


def _bench(*argv, env=None, timeout=600):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_without_gpus_fails_loudly():
    """`bench.py --gpus 8` must start 8 ranks or fail: it never reports one rank as 8 GPUs."""
    import torch
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("8 GPUs visible: the run would be real")
    r = _bench("--gpus", "8", "--steps", "2", "--warmup", "1")
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout
    assert "--gpus 8" in r.stderr


def test_bench_refuses_world_size_mismatch():
    """Under torch.distributed.run WORLD_SIZE must equal --gpus."""
    r = _bench("--gpus", "4", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and "n_gpus" not in r.stdout


def test_launcher_rehearsal_with_eight_ranks():
    """The N = 8 plumbing, rehearsed without GPUs (a GPU box admits six processes on its card, so eight ranks cannot share one):
    `bench.py --gpus 8 --rehearse-launcher` starts eight ranks itself (torch.distributed.run as a child process), they
    rendezvous over gloo, pin themselves to pairwise disjoint CPU sets, run the barriers, the MAX all-reduce and the
    all-gather of the per-clip records, and rank 0 writes the 8-entry concat list (concat.sh:248 / join.ts:51-53 is the
    model).  A functional test: the line says that nothing was measured."""
    import json
    r = _bench("--gpus", "8", "--rehearse-launcher", "--steps", "2", "--batch", "8")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["records"] == 8 and line["concat_list_lines"] == 8 and line["config"]["clips"] == 8
    assert line["value"] is None and "unmeasured" in line["scaling"] and "nothing was measured" in line["rehearsal"]
    sets = []
    for txt in line["rank_cpus_all"]:
        first, last = (int(v) for v in txt.split(" ")[0].split("-"))
        sets.append(set(range(first, last + 1)))
    assert len(sets) == 8
    if len(os.sched_getaffinity(0)) >= 8:                     # enough CPUs for eight shares: they do not overlap
        assert all(not (a & b) for i, a in enumerate(sets) for b in sets[i + 1:]), line["rank_cpus_all"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    listing = open(os.path.join(root, "gpurun_out", "concat_list_8gpu_rehearsal.txt")).read().splitlines()
    assert listing == [f"file 'clip_{i:02d}_stabilised.mp4'" for i in range(8)]


def test_bench_cpu_baseline_legs_of_the_plane_wise_outputs():
    """bench.py's CPU-baseline leg for --out-format nv12-planar / p010-planar times the oracle's plane-wise restatement (createMap + remap of the two
    planes), on a bounded sample: small frames here, a fraction of a second each."""
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import oracle
    import synth
    w, h = 320, 180
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    f = synth.nv12(0, w, h)
    for frame, depth, blend in ((f, 8, 0), (f.astype(np.uint16) << 8, 10, 1)):
        leg = bench.cpu_baseline_planar(frame, w, h, K, Ko, cw, ch, depth, blend=blend, budget_s=0.2, threads=1)
        assert leg["kind"] == "port" and leg["unit"] == "frames/s" and leg["cores"] == 1 and leg["value"] > 0
        assert "plane-wise" in leg["sample"] and ("NV12" if depth == 8 else "P010") in leg["sample"]
