"""Clip-level sharding across GPUs.

The path shards only at clip granularity (frames of a clip are sequentially dependent:
FrameSourceWarp.cpp:422-427, 441-444), so the reference's own parallelism is a process fan-out
over clips (concat.sh:200-201,248 `xargs -P`).  Here: one process per GPU, clip i -> rank
i mod N, no per-frame communication.  The only collective is one all-gather of a fixed-size
per-rank record at the end of the run (RCCL over xGMI on GPUs, gloo on CPU), which doubles as the
barrier before rank 0 writes the ffmpeg concat list (format of join.ts:51-53 / concat.sh:19-23).
"""
import zlib

import numpy as np

# cpu_first / cpu_last / cpu_count: the CPUs the rank ran on; cpu_us: user + system time of the rank's process over its timed region (0 if not given)
RECORD_FIELDS = ("rank", "clip", "frames", "elapsed_ns", "crc", "cpu_first", "cpu_last", "cpu_count", "cpu_us")


def assign_clips(n_clips, world_size):
    """clip i -> rank i mod N; returns the list of clip ids per rank."""
    return [list(range(r, n_clips, world_size)) for r in range(world_size)]


def crc_of(array_u8):
    """Checksum of an output frame (host numpy uint8) for the result record."""
    return zlib.crc32(np.ascontiguousarray(array_u8).tobytes()) & 0xFFFFFFFF


def gather_records(records, device=None):
    """All-gather the per-clip records of every rank.  `records`: list of dicts with RECORD_FIELDS
    for the clips this rank processed.  Every rank returns the full, clip-ordered list.  This is the
    run's single collective; it also acts as the final barrier."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    if not dist.is_initialized():
        return sorted(records, key=lambda r: r["clip"])
    n_local = torch.tensor([len(records)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    cap = int(max(int(c.item()) for c in counts))
    buf = torch.zeros((max(cap, 1), len(RECORD_FIELDS)), dtype=torch.int64, device=device)
    for i, r in enumerate(records):
        buf[i] = torch.tensor([int(r.get(k, 0)) for k in RECORD_FIELDS], dtype=torch.int64)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    allrec = []
    for rank, (t, c) in enumerate(zip(out, counts)):
        for row in t[: int(c.item())].cpu().tolist():
            allrec.append(dict(zip(RECORD_FIELDS, row)))
    return sorted(allrec, key=lambda r: r["clip"])


def concat_list(paths):
    """ffmpeg concat-demuxer list: one `file '<path>'` line per clip, in clip order."""
    return "".join("file '%s'\n" % p.replace("'", "'\\''") for p in paths)
