"""UAT 978 rate on one GPU: device-resident synthetic stream, process_buffer semantics over the whole stream.

    python tools/uat_rate.py [--mib 1024] [--reps 5] [--cpu-mib 64]

Prints one JSON line: kernel times (HIP events inside the library), wall time including the host resolve, frames found,
and the CPU oracle's rate on a bounded sample of the same stream.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import libadsb_amd as A  # noqa: E402
from libadsb_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-mib", type=int, default=64)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cfg", default="", help="generator overrides, e.g. pct_corrupt=0,pct_uplink=0")
    a = ap.parse_args()
    piece = 64 << 20
    npieces = max(1, (a.mib << 20) // piece)
    over = {k: int(v) for k, v in (kv.split("=") for kv in a.cfg.split(",") if kv)}
    cfg = synth.default_cfg978(**over)
    dev = torch.empty(npieces * piece, dtype=torch.uint8, device="cuda")
    first = None
    for k in range(npieces):
        h = synth.fill978(k, piece, cfg)
        if first is None:
            first = h
        dev[k * piece:(k + 1) * piece].copy_(torch.from_numpy(h))
    torch.cuda.synchronize()
    nsamples = dev.numel() // 2
    u = A.Uat978()
    frames, done = u.process_device(dev.data_ptr(), nsamples)  # warm-up, sizes the scratch
    walls, scans, demods = [], [], []
    for _ in range(a.reps):
        t = time.perf_counter()
        frames, done = u.process_device(dev.data_ptr(), nsamples, collect=False)
        walls.append(time.perf_counter() - t)
        tm = u.timing()
        scans.append(tm["scan_ms"])
        demods.append(tm["demod_ms"])
    frames, done = u.process_device(dev.data_ptr(), nsamples)
    out = {
        "workload": "UAT 978 synthetic u8 IQ, %d MiB device resident, one process_buffer over the stream" % (npieces * 64),
        "samples": nsamples, "frames": len(frames), "consumed": done,
        "scan_kernels_ms": min(scans), "demod_kernel_ms": min(demods), "wall_ms": min(walls) * 1e3,
        "msamples_per_s_kernels": nsamples / ((min(scans) + min(demods)) * 1e-3) / 1e6,
        "msamples_per_s_wall": nsamples / min(walls) / 1e6,
        "hbm_read_gbs_sign_kernel_algorithmic": 2 * nsamples / (min(scans) * 1e-3) / 1e9,
        "candidates_per_run": tm["candidates"] // (a.reps + 2), "extra_lookups": tm["extra_lookups"], "host_wall_ms": tm["host_wall_ms"],
    }
    if not a.no_cpu:
        from oracle import oracle_py as O
        n = min(a.cpu_mib << 20, first.size)
        phi = O.phase_lut978()[first[:n].view(np.uint16)]
        t = time.perf_counter()
        want, _ = O.process_buffer978(phi)
        dt = time.perf_counter() - t
        out["cpu_oracle"] = {"msamples_per_s": n / 2 / dt / 1e6, "cores": 1, "sample": "%d MiB of the same stream (LUT map excluded)" % (n >> 20),
                             "frames": len(want)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
