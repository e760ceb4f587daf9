#!/usr/bin/env python3
"""bench.py -- Msamples/s of the 1090ES IQ -> Mode S record path on MI355X.

One "step" = one pass of the hot path over one batch: every rank demodulates its own shard of
reference buffers (default 1 GiB = 4096 buffers of 262144 B per GPU, already resident in HBM) and
brings the sorted candidate records back to the host.  Steps are pipelined over two result slots,
so the record copy of step k overlaps the kernels of step k+1; the timed region contains K complete
steps (all records on the host).  No collective is on the data path (independent buffers, SURVEY.md
section 8e); ranks only agree on the elapsed time (max) and sum their record counts.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SETUP_STEPS = 40  # untimed, before the warm-up steps: first-touch of the result regions, clock ramp
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def measured_traffic(bytes_per_gpu):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic.json:
    FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate --pmc runs of this same workload); None when no matching entry."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        e = t.get(str(bytes_per_gpu))
        return int(e["traffic_bytes"]) if e else None
    except Exception:
        return None


def cpu_baseline(iq_host, nbuf_sample, buffer_bytes):
    """The CPU restatement (oracle) timed on this host: 1 thread over a bounded sample of the same buffers."""
    from oracle import oracle_py as O
    o = O.Oracle1090()
    lib = O.lib()
    t0 = time.perf_counter()
    for b in range(nbuf_sample):
        chunk = iq_host[b * buffer_bytes:(b + 1) * buffer_bytes]
        lib.oracle1090_handle_data(o._h, chunk.ctypes.data, chunk.size, None, None)
    dt = time.perf_counter() - t0
    st = o.stats()
    samples = nbuf_sample * buffer_bytes // 2
    out = {"value": round(samples / dt / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "%d of the same reference buffers (%d MiB), one thread, oracle/liboracle1090.so (gcc -O3 -march=x86-64-v3), %.2f s"
                     % (nbuf_sample, nbuf_sample * buffer_bytes >> 20, dt),
           "accepted_frames": int(st["accepted"])}
    # all host cores: buffers sharded over threads (ctypes releases the GIL), one handler state per thread
    try:
        from concurrent.futures import ThreadPoolExecutor
        ncores = len(os.sched_getaffinity(0))
        if ncores > 1:
            def work(tid):
                oo = O.Oracle1090()
                for b in range(tid, nbuf_sample, ncores):
                    c = iq_host[b * buffer_bytes:(b + 1) * buffer_bytes]
                    lib.oracle1090_handle_data(oo._h, c.ctypes.data, c.size, None, None)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(ncores) as ex:
                list(ex.map(work, range(ncores)))
            dt2 = time.perf_counter() - t0
            out["all_cores"] = {"value": round(samples / dt2 / 1e6, 2), "cores": ncores}
    except Exception as e:  # the single-thread figure is the baseline; this is extra
        out["all_cores"] = {"error": str(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps; the clocks and the two-slot pipeline need a few dozen steps to settle (20 steps read ~5 %% slower than 200 or 2000)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mib", type=int, default=1024, help="MiB of u8 IQ per GPU (weak scaling)")
    ap.add_argument("--cpu-buffers", type=int, default=4096, help="reference buffers in the CPU-baseline sample (0: skip)")
    ap.add_argument("--serial", action="store_true", help="submit/fetch one step at a time (no overlap of the record copy with the next "
                    "scan); used for rocprofv3 runs, where the runtime's shader-based copy would otherwise co-run with the scan kernel")
    ap.add_argument("--workload", choices=["1090", "uat978"], default="1090", help="1090: the headline metric (BASELINE configs[1]+[2]); "
                    "uat978: BASELINE configs[4], one independent stream per GPU (the UAT path does not shard: replicas only)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="N > 1 ranks all on cuda:0 with the gloo backend: exercises the "
                    "multi-rank code path (sharding, barrier, MAX/SUM reductions) on a one-GPU box; the numbers mean nothing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: libadsb_amd has no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import libadsb_amd as A
    from libadsb_amd import synth

    if args.workload == "uat978":
        return main_uat978(args, rank, local_rank, world, dist, A, synth)

    BB = A.REF_BUFFER_BYTES
    nbuf = (args.mib << 20) // BB
    first = rank * nbuf  # rank r owns buffers [r*nbuf, (r+1)*nbuf) of the recording
    ncpu = max(1, len(os.sched_getaffinity(0)) // max(1, world))
    iq_host, injected = synth.fill_range(first, nbuf, nthreads=ncpu)
    d_iq = torch.from_numpy(iq_host).cuda()
    torch.cuda.synchronize()

    sc = A.Scanner(local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    nbytes = d_iq.numel()

    def run(steps):
        """`steps` pipelined steps; returns (records of the last step, sum of scan-kernel ms, sum of enqueue-to-count ms)."""
        k_ms = t_ms = 0.0
        rec = None
        if args.serial:
            for i in range(steps):
                sc.submit(d_iq.data_ptr(), nbytes, BB, stream, 0)
                rec = sc.fetch(0, copy=False)
                a, b = sc.timing(0)
                k_ms += a
                t_ms += b
            return rec, k_ms, t_ms
        sc.submit(d_iq.data_ptr(), nbytes, BB, stream, 0)
        for i in range(1, steps):
            sc.submit(d_iq.data_ptr(), nbytes, BB, stream, i & 1)
            rec = sc.fetch((i - 1) & 1, copy=False)
            a, b = sc.timing((i - 1) & 1)
            k_ms += a
            t_ms += b
        rec = sc.fetch((steps - 1) & 1, copy=False)
        a, b = sc.timing((steps - 1) & 1)
        return rec, k_ms + a, t_ms + b

    # Set-up, not measurement: the first scans fault in the record regions (512 MiB of address space, touched where
    # used) and the clocks ramp up from idle; both are one-off costs of a long-running demodulator.  SETUP_STEPS untimed
    # steps absorb them before the W warm-up steps the caller asked for (reported as "setup_steps").
    run(SETUP_STEPS)
    if args.warmup > 0:
        run(args.warmup)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    rec, k_ms, t_ms = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0

    nrec = int(len(rec))
    rec = rec.copy()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([nrec, injected], dtype=torch.int64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        nrec_all, injected_all = int(c[0].item()), int(c[1].item())
    else:
        nrec_all, injected_all = nrec, injected

    if rank == 0:
        samples_rank = nbytes // 2
        samples_all = samples_rank * world
        value = samples_all * args.steps / elapsed / 1e6
        kernel_ms = k_ms / args.steps
        alg_bytes = 2.0 * samples_rank + 32.0 * nrec
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # host half on the records of one step (rank 0's shard): accepted frames -> msgs/s
        res = A.Resolver()
        t1 = time.perf_counter()
        accepted, _, _ = res.feed(rec, BB // 2, nbuf, collect=False)
        resolve_s = time.perf_counter() - t1
        out = {
            "metric": "Msamples/s demodulated (1090ES u8 IQ -> Mode S frame records)",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "setup_steps": SETUP_STEPS, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 in / u16 integer (bit-exact)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]+[2]: %d MiB synthetic u8 IQ per GPU (%d reference buffers of 262144 B, splitmix64 seed "
                                   "0x1090AD5B, noise +-3, ~1 frame / 2000 samples), fused magnitude + preamble gates + Manchester slice + "
                                   "phase retry + CRC-24 + 1-bit repair; reference demodulates 2 samples/us (2.0 MS/s, SURVEY.md F5)"
                                   % (args.mib, nbuf),
                       "bytes_per_gpu": nbytes, "buffers_per_gpu": nbuf, "sharding": "contiguous buffer ranges, no data-path collective",
                       "pipelined": not args.serial},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(nbytes),
                         "kernel": "scan1090_kernel", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes": int(alg_bytes)},
            "records_per_step": nrec_all, "frames_injected": injected_all,
            "decoded_msgs_per_step_rank0": int(accepted),
            # frames through the sequential host half (ICAO cache, decode, CPR, aircraft state): the GPU hands over records for
            # `accepted` frames per step in ms_per_step, the host resolves them in host_resolve_ms on one core; a pipeline of the
            # two sustains the slower of the two rates (the host's)
            "decoded_msgs_per_s": round(min(accepted * world * args.steps / elapsed, accepted * world / max(resolve_s, 1e-9)), 1),
            "decoded_msgs_per_s_gpu_side": round(accepted * world * args.steps / elapsed, 1),
            "gpu_enqueue_to_count_ms": round(t_ms / args.steps, 4),
            "host_resolve_ms_rank0": round(resolve_s * 1e3, 2),
        }
        if world == 1 and args.cpu_buffers > 0:
            out["cpu_baseline"] = cpu_baseline(iq_host, min(args.cpu_buffers, nbuf), BB)
        print(json.dumps(out), flush=True)
    sc.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_uat978(args, rank, local_rank, world, dist, A, synth):
    """BASELINE configs[4]: UAT 978 u8 IQ (the reference's UAT input is u8, not i16) -> frames.  A step is one
    process_buffer over the rank's whole device-resident stream: discriminator + sync search, sync re-check + slicing +
    Reed-Solomon per match, records to the host, the host scan loop (no up-calls).  Streams are independent replicas."""
    piece = 64 << 20
    npieces = max(1, (args.mib << 20) // piece)
    cfg = synth.default_cfg978()
    dev = torch.empty(npieces * piece, dtype=torch.uint8, device="cuda")
    first = None
    for k in range(npieces):
        h = synth.fill978(rank * npieces + k, piece, cfg)
        first = h if first is None else first
        dev[k * piece:(k + 1) * piece].copy_(torch.from_numpy(h))
    torch.cuda.synchronize()
    nsamples = dev.numel() // 2
    u = A.Uat978(local_rank)

    def run(steps):
        scan = demod = 0.0
        frames = 0
        for _ in range(steps):
            out, _ = u.process_device(dev.data_ptr(), nsamples, collect=False)
            tm = u.timing()
            scan += tm["scan_ms"]
            demod += tm["demod_ms"]
        return scan, demod

    frames, consumed = u.process_device(dev.data_ptr(), nsamples)  # sizes the scratch; frame count for the report
    if args.warmup > 0:
        run(args.warmup)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    matches0 = u.timing()["candidates"]
    t0 = time.perf_counter()
    scan_ms, demod_ms = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    matches = (u.timing()["candidates"] - matches0) // args.steps
    nframes = len(frames)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([nframes], dtype=torch.int64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        nframes_all = int(c.item())
    else:
        nframes_all = nframes
    if rank == 0:
        kernel_ms = scan_ms / args.steps
        alg_bytes = 2.0 * nsamples
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        tm = u.timing()
        out = {
            "metric": "Msamples/s demodulated (UAT 978 u8 IQ -> Reed-Solomon-corrected frames)",
            "value": round(nsamples * world * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 in / u16 phase, GF(256) (bit-exact vs oracle)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: %d MiB synthetic UAT 978 u8 IQ per GPU (CPFSK h=0.6, 2 samples/bit, seed 0x978AD5B), "
                                   "phase LUT + discriminator + 18-bit sync search + 36-bit sync re-check + slicing + RS(30,18)/RS(48,34)/"
                                   "6xRS(92,72) on the GPU, dump978 scan-loop rules on the host; parity unpinned (dump978 is un-vendored)"
                                   % (npieces * 64),
                       "bytes_per_gpu": int(dev.numel()), "sharding": "replicas only: one independent stream per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic("uat978:%d" % dev.numel()), "kernel": "uat_scan_iq_kernel",
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes": int(alg_bytes)},
            "frames_per_step": nframes_all, "demod_kernel_ms": round(demod_ms / args.steps, 4), "matches_per_step_rank0": int(matches),
            "host_wall_ms_last_step": tm["host_wall_ms"],
        }
        if world == 1 and args.cpu_buffers > 0:
            from oracle import oracle_py as O
            import numpy as np
            phi = O.phase_lut978()[first.view(np.uint16)]
            t1 = time.perf_counter()
            want, _ = O.process_buffer978(phi)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(first.size / 2 / dt / 1e6, 1), "unit": "Msamples/s", "cores": 1, "kind": "port",
                                   "sample": "oracle978 process_buffer over the first 64 MiB of the same stream (%.2f s), LUT map excluded; %d frames"
                                             % (dt, len(want))}
        print(json.dumps(out), flush=True)
    u.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
