"""Multi-GPU use of the scanner: independent reference buffers are split over ranks, records come back to one rank.

Every 262144-byte reference buffer is demodulated independently, so a recording is partitioned at buffer
granularity with no halo and no exchange on the data path.  The only communication is the gather of the (small) record
arrays to the rank that runs the sequential resolver: an all_gather of the per-rank counts followed by an all_gather of
the padded record bytes -- RCCL over xGMI with backend "nccl", gloo in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import RECORD_DTYPE


def shard_range(nbuf_total, rank, world):
    """Contiguous block of buffers owned by `rank`: [first, first + count)."""
    first = rank * nbuf_total // world
    last = (rank + 1) * nbuf_total // world
    return first, last - first


def gather_records(records, first_buffer, group=None, device=None):
    """All ranks call this with their local records (buffer indices local to the shard) and the global index of the
    shard's first buffer.  Every rank receives the concatenation in rank (= stream) order with global buffer indices."""
    world = dist.get_world_size(group)
    device = device or (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    rec = np.ascontiguousarray(records, dtype=RECORD_DTYPE).copy()
    rec["buffer"] += first_buffer
    count = torch.tensor([len(rec)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    payload = torch.zeros(cap * RECORD_DTYPE.itemsize, dtype=torch.uint8)
    if len(rec):
        payload[:len(rec) * RECORD_DTYPE.itemsize] = torch.from_numpy(rec.view(np.uint8).reshape(-1))
    payload = payload.to(device)
    parts = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(parts, payload, group=group)
    out = [np.frombuffer(p.cpu().numpy().tobytes()[:n * RECORD_DTYPE.itemsize], dtype=RECORD_DTYPE) for p, n in zip(parts, counts)]
    return np.concatenate(out) if out else np.zeros(0, RECORD_DTYPE)
