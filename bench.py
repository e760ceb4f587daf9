#!/usr/bin/env python3
"""bench.py -- Msamples/s of the 1090ES IQ -> Mode S record path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1 (BASELINE configs[1]+[2]): one "step" = one pass of the hot path over one batch: the GPU demodulates 1 GiB of
synthetic u8 IQ (4096 reference buffers of 262144 B, already resident in HBM) and brings the sorted candidate records
(packed: record head + GPU-decoded fields, 32 bytes each) back to the host.  Steps are pipelined over two result slots (the
record copy of step k overlaps the kernels of step k+1); the timed region contains K complete steps, after W warm-up steps
and SETUP_STEPS untimed set-up scans (the first scans after an idle period run slow, profiles/r03_sweep.txt).  The same JSON
line carries the roofline of the dominant kernel, the CPU baseline, an "end_to_end" block (records -> callbacks on the host,
host-resident input including the upload), the 2.4 MS/s mode ("mode_2400") and the UAT 978 workload (configs[4], "uat978").

N > 1 (BASELINE configs[3], the recorded-file case): the recording is N GiB (weak scaling: 1 GiB per GPU), rank r owns
buffers [r*B/N, (r+1)*B/N) (no halo: buffers are independent; a trailing partial buffer is never
delivered, reference RTLSDR.hpp:419-442).  One step = every rank scans its shard and its packed records land in rank 0's
host memory, in recording order: each GPU writes its own segment of node-shared page-locked memory over its own PCIe link,
and a 32-byte header per rank goes through the shared control page (shard.NodeGather: no collective per step; --rccl-gather
or a node without /dev/shm room: the records themselves are gathered over RCCL).  Steps are pipelined the same way.  `value` is the GPU side; a second timed loop
with the sequential resolver inside gives end_to_end_msamples_per_s.

--workload uat978: BASELINE configs[4]; with N > 1 replicas (`value`) and one N GiB stream cut over the ranks.

Started bare with --gpus N > 1 (no RANK in the environment) the script starts its N ranks itself as child processes
(python -m torch.distributed.run, rendezvous on 127.0.0.1) before anything touches the GPU and exits with their code.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SETUP_STEPS = 400  # untimed, before the warm-up steps: first-touch of the result regions, and the ~100-250 scans after an idle period that run 3-4 % slow (profiles/r03_sweep.txt)
UAT_SETUP_STEPS = 60  # the same for the UAT workload (serial calls of 0.75 ms)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps; the clocks and the two-slot pipeline need a few dozen steps to settle (20 steps read ~5 %% slower than 200 or 2000)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mib", type=int, default=1024, help="MiB of u8 IQ per GPU (weak scaling)")
    ap.add_argument("--cpu-buffers", type=int, default=4096, help="reference buffers in the CPU-baseline sample (0: skip)")
    ap.add_argument("--sharded-step-on-one-rank", action="store_true", help="run the N > 1 step (scan, record hand-over, nccl) with a world of one, "
                    "started under torch.distributed.run --nproc-per-node 1: the device path of the hand-over on a box with one GPU")
    ap.add_argument("--rccl-gather", action="store_true", help="N > 1: gather the records themselves over RCCL to rank 0's GPU and copy them to its "
                    "host from there (the round-2 first form), instead of every GPU writing into node-shared page-locked host memory")
    ap.add_argument("--serial", action="store_true", help="submit/fetch one step at a time (no overlap of the record copy with the next "
                    "scan); used for rocprofv3 runs, where the runtime's shader-based copy would otherwise co-run with the scan kernel")
    ap.add_argument("--workload", choices=["1090", "uat978"], default="1090", help="1090: the headline metric (BASELINE configs[1]+[2], and "
                    "configs[3] when --gpus > 1); uat978: BASELINE configs[4] alone, one independent stream per GPU (replicas only)")
    ap.add_argument("--rate", type=int, choices=[20, 24], default=20, help="samples per microsecond x 10.  20: the reference's demodulator "
                    "(parity-green, the headline).  24: the library's own 2.4 MS/s mode (ADSB_AMD_MODE_2400) on the same pulse trains sampled at "
                    "2.4 MS/s -- BASELINE.json quotes that rate, nothing in the reference demodulates it (SURVEY.md F3/F5): parity unpinned, "
                    "the kernel is checked against its specification oracle/oracle2400.c")
    ap.add_argument("--time-every", type=int, default=4, help="HIP events around the scan kernel on every n-th launch of the timed region (1: all)")
    ap.add_argument("--noise", type=int, default=None, help="N = 1: background noise amplitude of the synthetic input (default 3, the BASELINE workload; "
                    "any other value is a sensitivity run, labelled as such: profiles/r05_sensitivity.txt)")
    ap.add_argument("--spacing", type=int, default=None, help="N = 1: mean frame start-to-start spacing in samples (default 2000)")
    ap.add_argument("--rotate", type=int, default=1, help="N = 1: K distinct device-resident inputs of --mib each (generator buffers k * nbuf ...), step i scans "
                    "input i mod K: nothing a step reads was read by the K - 1 steps before it, so neither the L2s nor the 256 MiB memory-side cache "
                    "can serve it (K = 1, the default, rescans one resident input; profiles/r06_rotate.txt holds the A/B)")
    ap.add_argument("--depth", type=int, default=2, help="N = 1: scans on the stream at any time in the pipelined loop (2, the default, or 3: the context has three "
                    "result slots; measured no faster, profiles/r06_ordering_in_kernel.txt)")
    ap.add_argument("--streams", type=int, default=1, help="N = 1: the scans of the pipelined loop alternate over this many streams (with 2 and --depth 3 a "
                    "scan kernel may start while the one before it drains; its ordering pass then rides in front of the kernel two steps on, the next one on its stream)")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the end_to_end and uat978 blocks (profiling runs)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="N > 1 ranks all on cuda:0 with the gloo backend: exercises the "
                    "multi-rank code path (sharding, record gather, resolve) on a one-GPU box; the numbers mean nothing")
    ap.add_argument("--dump-stream", default=None, help="N > 1: rank 0 writes the gathered records and the resolved callback stream (frames, "
                    "aircraft snapshots) of the last step to this .npz file, for the parity test (small --mib only)")
    return ap.parse_args(argv)


def self_launch(args):
    """Started bare with --gpus N > 1: start the N ranks as fresh child processes and exit with their return code.  Nothing in
    this (parent) process has touched the GPU, and it never replaces itself: it waits."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def measured_traffic(key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/traffic_from_profile.py from the tracked summary: FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate --pmc runs of this
    same workload); None when there is no entry for this workload size."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        e = t.get(str(key))
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


WORKLOAD_1090 = ("%d MiB synthetic u8 IQ per GPU (%d reference buffers of 262144 B, splitmix64 seed 0x1090AD5B, noise +-3, ~1 frame / 2000 "
                 "samples), fused magnitude + preamble gates + Manchester slice + phase retry + CRC-24 + 1-bit repair; reference demodulates "
                 "2 samples/us (2.0 MS/s, SURVEY.md F5)")


def bind_near_gpu(torch, local_rank):
    """Best effort, N > 1 only: keep this rank's threads on the CPUs of the NUMA node its GPU hangs off, so that the page-locked memory the
    rank allocates -- its segments of the node-shared hand-over area among it -- is local to the GPU's PCIe root.  Returns the node or None."""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        dev = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % dev).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            # every thread this process has by now (the HIP runtime's among them: sched_setaffinity(0, ..) moves the calling thread only);
            # threads started later inherit the mask of the thread that starts them
            for tid in os.listdir("/proc/self/task"):
                try:
                    os.sched_setaffinity(int(tid), cpus)
                except OSError:
                    pass
            return node
    except Exception:
        pass
    return None


def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import numpy as np  # noqa: F401
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: libadsb_amd has no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    args.numa_node = None
    if world > 1 and not args.rehearse_on_one_gpu:
        args.numa_node = bind_near_gpu(torch, local_rank)
    dist = None
    if world > 1 or args.sharded_step_on_one_rank:
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import libadsb_amd as A
    from libadsb_amd import synth

    if args.workload == "uat978":
        out = bench_uat978(args, rank, local_rank, world, dist, A, synth, torch)
        if world > 1:
            one = bench_uat978_one_stream(args, rank, local_rank, world, dist, A, synth, torch)
            if rank == 0:
                out["one_stream_cut_over_ranks"] = one
        if rank == 0:
            print(json.dumps(out), flush=True)
    elif world == 1 and not args.sharded_step_on_one_rank:
        out = bench_1090_single(args, local_rank, A, synth, torch)
        if not args.no_extras:
            try:
                u = bench_uat978(argparse.Namespace(**{**vars(args), "steps": max(50, min(args.steps, 100)), "warmup": min(args.warmup, 3)}),  # (its own step count, reported in the block: a window of calls in flight starts and ends with an empty pipeline, 0.75 ms for its first call)
                                 0, local_rank, 1, None, A, synth, torch)
                out["uat978"] = {k: u[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config", "roofline", "path_roofline", "frames_per_step",
                                                   "demod_kernel_ms", "dominant_kernel", "matches_per_step_rank0", "pipelined", "ms_per_step_serial", "timing",
                                                   "host_wall_ms_last_step", "cpu_baseline") if k in u}
            except Exception as e:  # the headline line must not be lost to the second workload
                out["uat978"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    else:
        try:
            out = bench_1090_sharded(args, rank, local_rank, world, dist, A, synth, torch)
        except BaseException as e:
            # a rank that fails leaves ONE JSON line of its own behind (every rank that fails does; rank 0's is not special here): what failed, and
            # which record transport each rank had ended on as far as this rank knows
            print(json.dumps({"metric": "Msamples/s demodulated (1090ES u8 IQ -> Mode S frame records)", "value": None, "n_gpus": world, "failed_rank": rank,
                              "error": repr(e)[:400], "record_transport_by_rank": getattr(args, "transport_by_rank", None),
                              "record_transport_this_rank": None if getattr(args, "transport_by_rank", None) is None else args.transport_by_rank[rank]}), flush=True)
            raise
        if rank == 0:
            print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def make_runner(args, sc, d_iq, BB, stream):
    inputs = d_iq if isinstance(d_iq, (list, tuple)) else [d_iq]  # --rotate K: step i scans inputs[i mod K]
    streams = [stream] + list(getattr(args, "more_streams", []))  # --streams S: step i is launched on stream i mod S
    nbytes = inputs[0].numel()
    seen = [0]  # steps submitted so far by this runner (the rotation goes on across calls of run)
    made = [0, 0]  # records delivered, steps delivered (the inputs of a rotation differ in their record counts)

    def ptr():
        p = inputs[seen[0] % len(inputs)].data_ptr()
        seen[0] += 1
        return p

    def run(steps):
        """`steps` pipelined steps, each delivering the sorted records with their decoded fields to the host in the packed hand-over
        form (adsb_amd_packed_t: 32 bytes a record); returns (packed records of the last step, sum of scan-kernel ms over the steps
        that carried timing events, how many did, sum of scan-start-to-count ms over the same steps)."""
        acc = [0.0, 0, 0.0]
        rec = None

        def note(slot):
            tm = sc.timing_if_timed(slot)
            if tm is not None:
                acc[0] += tm[0]
                acc[1] += 1
                acc[2] += tm[1]
        def took(r):
            made[0] += len(r)
            made[1] += 1
            return r
        if args.serial:
            for i in range(steps):
                sc.submit(ptr(), nbytes, BB, stream, 0)
                rec = took(sc.fetch_packed(0, copy=False))
                note(0)
            return rec, acc[0], acc[1], acc[2]
        if not sc.has_split_fetch():  # (an older build under ADSB_AMD_LIB: tools/ab.py)
            sc.submit(ptr(), nbytes, BB, stream, 0)
            for i in range(1, steps):
                sc.submit(ptr(), nbytes, BB, stream, i & 1)
                rec = took(sc.fetch_packed((i - 1) & 1, copy=False))
                note((i - 1) & 1)
            rec = took(sc.fetch_packed((steps - 1) & 1, copy=False))
            note((steps - 1) & 1)
            return rec, acc[0], acc[1], acc[2]
        # `depth` scans on the stream at any time.  A step's ordering pass rides in front of the scan kernel after it (gather1090.hip.h), so its count
        # reaches the host some tens of microseconds into that kernel.  A step's records: wait for its count, start the copy, submit the slot's next
        # scan beside the copy (a scan writes the slot's raw regions only), then wait for the copy.
        depth = max(1, min(3, getattr(args, "depth", 2)))
        for k in range(min(depth, steps)):
            sc.submit(ptr(), nbytes, BB, streams[k % len(streams)], k)
        for i in range(steps):
            sc.fetch_packed_begin(i % depth)
            note(i % depth)
            if i + depth < steps:
                sc.submit(ptr(), nbytes, BB, streams[(i + depth) % len(streams)], i % depth)
            rec = took(sc.fetch_packed_end(i % depth, copy=False))
        return rec, acc[0], acc[1], acc[2]
    run.delivered = made
    return run


def bench_1090_single(args, local_rank, A, synth, torch):
    import numpy as np
    BB = A.REF_BUFFER_BYTES
    nbuf = (args.mib << 20) // BB
    ncpu = max(1, len(os.sched_getaffinity(0)))
    over = {k: v for k, v in (("noise_amp", args.noise), ("mean_spacing", args.spacing)) if v is not None}
    iq_host, injected = synth.fill_range(0, nbuf, nthreads=ncpu, rate_x10=args.rate, cfg=synth.default_cfg(**over) if over else None)
    if over:
        args.no_extras = True  # the other blocks are defined on the BASELINE workload
        args.cpu_buffers = 0
    d_iq = torch.from_numpy(iq_host).cuda()
    rotation = [d_iq]
    for k in range(1, max(1, args.rotate)):  # --rotate K: K - 1 further inputs of the same generator, the buffers behind the first nbuf
        more, _ = synth.fill_range(k * nbuf, nbuf, nthreads=ncpu, rate_x10=args.rate, cfg=synth.default_cfg(**over) if over else None)
        rotation.append(torch.from_numpy(more).cuda())
        del more
    if args.rotate > 1:
        args.no_extras = True  # an A/B of the headline kernel, not the headline line
        args.cpu_buffers = 0
    torch.cuda.synchronize()
    sc = A.Scanner(local_rank, mode=args.rate)
    sc.set_outputs(A.OUT_PACKED)  # the throughput path hands over record head + decoded fields, 32 bytes, no message bytes
    stream = torch.cuda.current_stream().cuda_stream
    nbytes = d_iq.numel()
    # (made here, before the first launch: which hardware queue a stream lands on follows the order in which the process makes its streams, and one made
    # late can share a queue with the scanner's copy stream or with the first stream -- no two kernels side by side then; DESIGN section 9.3 has the UAT case)
    more_streams = [torch.cuda.Stream() for _ in range(max(2, args.streams) - 1)]
    side_stream = more_streams[0]
    args.more_streams = [t.cuda_stream for t in more_streams[:max(1, args.streams) - 1]]
    run = make_runner(args, sc, rotation, BB, stream)
    if args.rate != 20:
        args.no_extras = True  # end-to-end and CPU legs are defined on the parity-green workload only
        args.cpu_buffers = 0

    # Set-up, not measurement: the first scans fault in the record regions (512 MiB of address space, touched where
    # used) and the clocks ramp up from idle; both are one-off costs of a long-running demodulator.  SETUP_STEPS untimed
    # steps absorb them before the W warm-up steps the caller asked for (reported as "setup_steps").
    sc.set_timing(1)
    _, k100, n100, _ = run(100)  # the first 100 of them with events on every scan: reported as kernel_ms_first_100
    run(SETUP_STEPS - 100)
    # In the timed region the two HIP events that take the scan kernel's start and end ride on every --time-every-th launch (they cost the
    # stream ~4 us per launch that carries them: profiles/r04_step_timeline.txt); kernel_ms is the mean over those launches.
    sc.set_timing(args.time_every)
    if args.warmup > 0:
        run(args.warmup)
    sc.set_timing(args.time_every)  # (counts from here: the first launch of the timed region carries events, then every n-th)
    torch.cuda.synchronize()
    run.delivered[0] = run.delivered[1] = 0
    t0 = time.perf_counter()
    rec, k_ms, k_n, t_ms = run(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if k_n == 0:
        raise SystemExit("--time-every %d leaves no timed launch among %d steps" % (args.time_every, args.steps))
    nrec = int(len(rec))
    nrec_mean = run.delivered[0] / max(1, run.delivered[1])  # == nrec unless --rotate K > 1 (the K inputs hold different frames)
    rec = rec.view(np.uint8).copy().view(rec.dtype)  # a byte copy (numpy copies a structured array field by field)

    # The scan kernel by itself, live: 24 scans one at a time (their ordering pass a launch of its own, no copy beside the kernel) with events on every
    # dispatch -- what rocprofv3 of `bench.py --serial` reads (profiles/), and what the kernel_ms of rounds 1-5 stood for before the pass moved in.
    scan_alone = None
    if not args.serial:
        sc.set_timing(1)
        alone = []
        for i in range(24):
            sc.submit(rotation[i % len(rotation)].data_ptr(), nbytes, BB, stream, 0)
            sc.fetch_packed(0, copy=False)
            tm = sc.timing_if_timed(0)
            if tm is not None and i >= 4:
                alone.append(tm[0])
        if alone:
            scan_alone = sum(alone) / len(alone)
    samples = nbytes // 2
    # The same loop with the scans alternating over TWO streams and three result slots in flight (--streams 2 --depth 3), once, after the timed region:
    # a scan kernel then starts while the one before it drains (its waves take the CUs' wave slots as the older kernel's persistent waves leave), and a
    # step's ordering pass rides in front of the kernel two steps on, the next one on its stream.  Reported beside the headline, not as it: events on
    # overlapping dispatches no longer read a kernel's own duration (kernel_ms would say 0.35 ms), and over 20 steps the longer drain -- results arrive a
    # kernel later -- takes back what the steady state gains (profiles/r06_two_streams.txt).
    overlapped = None
    if not args.serial and sc.has_split_fetch() and args.streams == 1 and not args.no_extras:
        args.more_streams, depth0 = [side_stream.cuda_stream], args.depth
        args.depth = 3
        run2 = make_runner(args, sc, rotation, BB, stream)
        sc.set_timing(0)
        run2(SETUP_STEPS)  # (the scans one at a time just before left the clocks where an idle period leaves them: kernel_ms_first_100)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rec2, _, _, _ = run2(100)
        torch.cuda.synchronize()
        sec2 = (time.perf_counter() - t2) / 100
        overlapped = {"streams": 2, "depth": 3, "steps": 100, "ms_per_step": round(sec2 * 1e3, 4), "msamples_per_s": round(samples / sec2 / 1e6, 1),
                      # (no kernel of this loop has a duration of its own -- they overlap --, but the chip is never idle: the whole step against the roofline)
                      "step_frac_of_hbm_peak": round((2.0 * samples + 32.0 * nrec) / sec2 / 1e9 / HBM_PEAK_GBS, 4),
                      "records_match_the_timed_loop": bool(len(rec2) == nrec and (len(rotation) > 1 or rec2.tobytes() == rec.tobytes())),
                      "what": "the headline loop with the scans alternating over two streams, three in flight: a kernel starts while the one before it drains; "
                              "steady state only (100 steps after %d untimed)" % SETUP_STEPS}
        args.more_streams, args.depth = [], depth0
    kernel_ms = k_ms / k_n
    alg_bytes = 2.0 * samples + 32.0 * nrec_mean
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    # host half on the records of one step: accepted frames -> msgs/s (no listener; with a native counting listener: end_to_end)
    # One resolver, the step's records fed six times as six consecutive stretches of the stream (a long-running handler: its helper
    # thread exists after the first large call); the frame count reported is the first stretch's, the time the best of the later five.
    resolve_s = 1e9
    res = A.Resolver(mode=args.rate, sample_clock_hz=100000 * args.rate)
    accepted, _, _ = res.feed(rec, BB // 2, nbuf, collect=False)
    for _ in range(5):
        t1 = time.perf_counter()
        res.feed(rec, BB // 2, nbuf, collect=False)
        resolve_s = min(resolve_s, time.perf_counter() - t1)
    res.close()
    # The replay as a whole, measured: the scan of step k + 1 runs on the GPU while this thread resolves the records of step k (one
    # resolver for the whole loop, as a long-running handler has).  This is the decoded-messages rate of a recorded-file job on one GPU;
    # it is bound by the sequential host half (host_resolve_ms per step), `value` above is the GPU side alone.
    pipe = None
    if not args.serial:
        resp = A.Resolver(mode=args.rate, sample_clock_hz=100000 * args.rate)
        psteps = max(4, min(args.steps, 40))
        torch.cuda.synchronize()
        tp = time.perf_counter()
        got = 0
        sc.submit(d_iq.data_ptr(), nbytes, BB, stream, 0)
        for i in range(1, psteps):
            sc.submit(d_iq.data_ptr(), nbytes, BB, stream, i & 1)
            got += resp.feed(sc.fetch_packed((i - 1) & 1, copy=False), BB // 2, nbuf, collect=False)[0]
        got += resp.feed(sc.fetch_packed((psteps - 1) & 1, copy=False), BB // 2, nbuf, collect=False)[0]
        ep = time.perf_counter() - tp
        resp.close()
        pipe = {"steps": psteps, "ms_per_step": round(ep / psteps * 1e3, 4), "msamples_per_s": round(samples * psteps / ep / 1e6, 1),
                "decoded_msgs_per_s": round(got / ep, 1), "decoded_msgs_per_step": round(got / psteps, 1),
                "what": "scan of step k+1 on the GPU under the host's resolve of step k (one thread + the resolver's helper), no listener"}
    out = {
        "metric": "Msamples/s demodulated (1090ES u8 IQ -> Mode S frame records)",
        "value": round(samples * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "setup_steps": SETUP_STEPS, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8 in / u16 integer (bit-exact)", "data": "synthetic",
        "config": {"workload": ("SENSITIVITY RUN, not the BASELINE workload (%s): " % ", ".join("%s=%d" % kv for kv in over.items()) if over else "") +
                               ("BASELINE configs[1]+[2]: " + WORKLOAD_1090 % (args.mib, nbuf)) if args.rate == 20 else
                               ("PARITY UNPINNED (no reference demodulates this rate, SURVEY.md F3/F5): %d MiB synthetic u8 IQ sampled at 2.4 MS/s (the "
                                "generator's pulse trains integrated over 1/2.4 us bins, frames at random sub-sample offsets) through the library's "
                                "2.4 MS/s mode: packed gate, five-phase preamble correlation, overlap-weighted Manchester slicing, CRC-24 + 1-bit "
                                "repair; kernel == oracle/oracle2400.c in the GPU tests" % args.mib),
                   "sample_rate_x10": args.rate,
                   "bytes_per_gpu": nbytes, "buffers_per_gpu": nbuf, "sharding": "one GPU", "pipelined": not args.serial,
                   "scans_on_the_stream": 1 if args.serial else max(1, min(3, args.depth)),
                   "rotate": {"inputs": len(rotation), "what": "step i scans device-resident input i mod K (K x %d MiB, generator buffers k * %d ...)" % (args.mib, nbuf),
                              "records_per_step_mean": round(nrec_mean, 1)} if len(rotation) > 1 else None},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(nbytes if args.rate == 20 else "mode2400:%d" % nbytes),
                     "kernel": "scan1090_kernel" if args.rate == 20 else "scan2400_kernel", "kernel_ms": round(kernel_ms, 4),
                     "kernel_ms_source": "HIP events on the kernel's dispatch (hipExtLaunchKernelGGL start/stop events, the stream the scan is launched on), "
                                         "%d of the %d timed launches (every %s), in the pipelined loop: since round 6 the kernel also does the ORDERING PASS of the "
                                         "scan before it (a quarter of its waves, in front of their own chunks: gather1090.hip.h; ~7 us of the figure), and the copy "
                                         "engine moves the previous step's records to the host beside it (2-5 %%) -- rocprofv3 of `bench.py --serial` (profiles/) has "
                                         "neither in or beside the scan kernel (the pass is a launch of its own there) and reads that much lower" % (k_n, args.steps, args.time_every),
                     "kernel_ms_first_100": round(k100 / max(1, n100), 4),
                     "scan_alone": None if scan_alone is None else {
                         "kernel_ms": round(scan_alone, 4), "frac": round((2.0 * samples + 32.0 * nrec_mean) / (scan_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "what": "the same kernel with nothing in front of it and no copy beside it: 20 scans one at a time after the timed region, HIP events on every "
                                 "dispatch (the figure rocprofv3 of `bench.py --serial` agrees with; rounds 1-5's kernel_ms was this kernel without the ordering pass)"},
                     "algorithmic_bytes": int(alg_bytes)},
        "records_per_step": nrec, "frames_injected": injected,
        "decoded_msgs_per_step_rank0": int(accepted),
        # frames through the sequential host half (ICAO cache, decode, CPR, aircraft state): measured in a loop that resolves step k on the
        # host while the GPU scans step k + 1 (`pipelined`); with --serial, the slower of the two separately timed halves
        "decoded_msgs_per_s": pipe["decoded_msgs_per_s"] if pipe else round(min(accepted * args.steps / elapsed, accepted / max(resolve_s, 1e-9)), 1),
        "pipelined": pipe,
        "overlapped": overlapped,
        "gpu_enqueue_to_count_ms": round(t_ms / k_n, 4),
        "host_resolve_ms_rank0": round(resolve_s * 1e3, 2),
        "host_resolve_note": "records + GPU-decoded fields -> ICAO gating, skip-ahead (helper thread) | batched CPR, aircraft update (calling thread), no listener; best of 5 stretches",
    }
    if not args.no_extras:
        out["end_to_end"] = end_to_end_1090(A, rec, iq_host, BB, nbuf, accepted)
        if pipe:
            out["end_to_end"]["pipelined_msamples_per_s"] = pipe["msamples_per_s"]
            out["end_to_end"]["pipelined_decoded_msgs_per_s"] = pipe["decoded_msgs_per_s"]
        # the "2.4 MS/s" flavour of the same configuration (BASELINE.json's wording): the library's own mode for that rate, see --rate
        try:
            iq24, inj24 = synth.fill_range(0, nbuf, nthreads=ncpu, rate_x10=24)
            d24 = torch.from_numpy(iq24).cuda()
            sc24 = A.Scanner(local_rank, mode=A.MODE_2400)
            sc24.set_outputs(A.OUT_PACKED)
            sc24.set_timing(args.time_every)
            run24 = make_runner(args, sc24, d24, BB, stream)
            run24(SETUP_STEPS)  # the GPU sat idle while the host generated this input: the same untimed set-up scans as the headline (see SETUP_STEPS)
            sc24.set_timing(args.time_every)
            torch.cuda.synchronize()
            t24 = time.perf_counter()
            r24, k24, n24, _ = run24(50)
            torch.cuda.synchronize()
            e24 = time.perf_counter() - t24
            k24 = k24 / n24 * 50  # (mean over the launches that carried events) x the 50 steps
            r24 = r24.view(np.uint8).copy().view(r24.dtype)
            res24 = A.Resolver(mode=A.MODE_2400, sample_clock_hz=2400000)
            acc24, _, _ = res24.feed(r24, BB // 2, nbuf, collect=False)
            out["mode_2400"] = {
                "value": round(samples * 50 / e24 / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(e24 / 50 * 1e3, 4), "kernel": "scan2400_kernel",
                "kernel_ms": round(k24 / 50, 4), "roofline_frac": round((2.0 * samples + 32.0 * len(r24)) / (k24 / 50 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "roofline": {"bound": "hbm", "achieved": round((2.0 * samples + 32.0 * len(r24)) / (k24 / 50 * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round((2.0 * samples + 32.0 * len(r24)) / (k24 / 50 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "traffic": measured_traffic("mode2400:%d" % nbytes), "algorithmic_bytes": int(2.0 * samples + 32.0 * len(r24))},
                "records_per_step": int(len(r24)), "frames_injected": int(inj24), "accepted_frames": int(acc24),
                "transmitted_frames_recovered": recovered_2400(A, synth, local_rank),
                "note": "PARITY UNPINNED: 1 GiB of the generator's pulse trains sampled at 2.4 MS/s through ADSB_AMD_MODE_2400 (specification "
                        "oracle/oracle2400.c; no reference demodulator for this rate exists, SURVEY.md F3/F5); kernel by part and what is left to do: DESIGN.md section 10, profiles/r06_mode2400_variants.txt"}
            sc24.close()
            del d24, iq24
        except Exception as e:
            out["mode_2400"] = {"error": repr(e)}
    if args.cpu_buffers > 0:
        out["cpu_baseline"] = cpu_baseline(iq_host, min(args.cpu_buffers, nbuf), BB)
    sc.close()
    del d_iq
    torch.cuda.empty_cache()
    return out


def recovered_2400(A, synth, local_rank, nbuf=16):
    """What the 2.4 MS/s mode brings back of what the generator transmits (the product on the GPU; no reference exists for this rate):
    fraction of the transmitted frames -- those that start at least 300 samples before their buffer's end -- decoded with the right
    bytes within one sample of where they start, at three settings.  The specification (oracle/oracle2400.c) is frozen; these three
    numbers in the driver's record are what would show a loss of sensitivity."""
    BB = A.REF_BUFFER_BYTES
    out = {}
    sc = A.Scanner(local_rank, mode=A.MODE_2400)
    for name, over in (("noise_3", {}), ("noise_10", {"noise_amp": 10}), ("noise_12_signals_20_60", {"noise_amp": 12, "amp_lo": 20, "amp_hi": 60})):
        cfg = synth.default_cfg(**over)
        total = hit = 0
        for b in range(nbuf):
            iq, frames = synth.fill(b, BB, cfg, manifest=True, rate_x10=24)
            got = {}
            for r in sc.scan(iq, BB):
                got.setdefault(bytes(r["msg"][:r["nbits"] // 8]), []).append(int(r["offset"]))
            for f in frames:
                if int(f["start"]) + 300 >= BB // 2:
                    continue
                msg = bytes(f["msg"][:int(f["nbits"]) // 8])
                cand = [msg]
                fb = int(f["flipped_bit"])
                if fb >= 0:  # transmitted with one bit flipped: a DF11/17 comes back repaired
                    m = bytearray(msg)
                    m[fb >> 3] ^= 0x80 >> (fb & 7)
                    cand.append(bytes(m))
                total += 1
                hit += any(c in got and any(abs(o - int(f["start"])) <= 1 for o in got[c]) for c in cand)
        out[name] = round(hit / max(1, total), 4)
        out[name + "_frames"] = total
    sc.close()
    return out


def end_to_end_1090(A, rec, iq_host, BB, nbuf, accepted):
    """What `value` leaves out (it is the device-resident demodulation rate): the host half with a listener attached, and the
    HandleData-shaped entry point on host memory, upload included.  Same 1 GiB, best of 3."""
    samples = nbuf * BB // 2
    out = {}
    best = 1e9
    for _ in range(3):
        r = A.Resolver()
        t = time.perf_counter()
        n, _, _ = r.feed(rec, BB // 2, nbuf, collect=False, count_callbacks=True)
        best = min(best, time.perf_counter() - t)
        r.close()
    out["records_to_callbacks_ms"] = round(best * 1e3, 3)
    out["callbacks"] = int(n)
    out["callbacks_per_s"] = round(n / best, 1)
    # HandleData over host memory: upload (pageable, then from page-locked memory) + scan + records to host + resolve + callbacks
    h = A.Handler1090()
    h.handle_data(iq_host[:64 * BB], BB, collect=False)
    best = 1e9
    for _ in range(3):
        hh = A.Handler1090()
        t = time.perf_counter()
        n2 = hh.handle_data(iq_host, BB, collect=False)
        best = min(best, time.perf_counter() - t)
        hh.close()
    assert n2 == accepted
    out["handle_data_pageable_1gib_ms"] = round(best * 1e3, 2)
    out["handle_data_pageable_msamples_per_s"] = round(samples / best / 1e6, 1)
    try:
        pin = A.PinnedBuffer(iq_host.size)
        pin.array[:] = iq_host
        best = 1e9
        for _ in range(3):
            hh = A.Handler1090()
            t = time.perf_counter()
            hh.handle_data(pin.array, BB, collect=False)
            best = min(best, time.perf_counter() - t)
            hh.close()
        out["handle_data_pinned_1gib_ms"] = round(best * 1e3, 2)
        out["handle_data_pinned_msamples_per_s"] = round(samples / best / 1e6, 1)
        # one live 262144-byte buffer from a page-locked ring slot (the transport's unit, RTLSDR.hpp:55)
        lat = []
        for k in range(200):
            t = time.perf_counter()
            h.handle_data(pin.array[(k % 64) * BB:(k % 64 + 1) * BB], 0, collect=False)
            lat.append(time.perf_counter() - t)
        lat.sort()
        out["live_buffer_ms_median"] = round(lat[len(lat) // 2] * 1e3, 4)
        out["live_buffer_ms_p99"] = round(lat[int(len(lat) * 0.99)] * 1e3, 4)
        pin.close()
    except Exception as e:
        out["pinned_error"] = repr(e)
    h.close()
    # recorded-file replay (RTLSDR.hpp:419-442 semantics) from the page cache: the batch path (mapped file, upload of batch k+1 beside
    # scan of k and resolve of k-1) over the whole GiB, and libadsb's own pacing -- the file through the 16-slot page-locked ring, one
    # HandleData per 262144-byte slot on the consumer thread -- over its first 256 MiB
    import tempfile
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(tmpdir, "adsb_amd_bench_%d.test.dat" % os.getpid())
    try:
        iq_host.tofile(path)
        best = 1e9
        for _ in range(3):
            hh = A.Handler1090()
            t = time.perf_counter()
            n3, _, _ = hh.replay_file(path, collect=False)
            best = min(best, time.perf_counter() - t)
            hh.close()
        assert n3 == accepted
        out["replay_file_1gib_ms"] = round(best * 1e3, 2)
        out["replay_file_gib_per_s"] = round(1.0 / best * (nbuf * BB / 2**30), 2)
        small = os.path.join(tmpdir, "adsb_amd_bench_%d.small.dat" % os.getpid())
        iq_host[:1024 * BB].tofile(small)
        hh = A.Handler1090()
        n4, _, _, nb, sec = hh.run_replay(small, collect=False)
        hh.close()
        os.unlink(small)
        out["ring_replay_buffers"] = int(nb)
        out["ring_replay_ms_per_buffer"] = round(sec / max(1, nb) * 1e3, 4)
        out["ring_replay_gib_per_s"] = round(nb * BB / 2**30 / sec, 2)
        out["ring_replay_x_realtime"] = round(nb * (BB // 2) / 2e6 / sec, 1)
    except Exception as e:
        out["replay_error"] = repr(e)
    finally:
        if os.path.exists(path):
            os.unlink(path)
    return out


def bench_1090_sharded(args, rank, local_rank, world, dist, A, synth, torch):
    """BASELINE configs[3]: the recording is world x args.mib, rank r scans buffers [r*B/N, (r+1)*B/N), records gathered on rank 0."""
    import numpy as np
    from libadsb_amd.shard import NodeGather, RootGather, Watchdog, shard_range
    BB = A.REF_BUFFER_BYTES
    nbuf_total = world * ((args.mib << 20) // BB)
    first, nbuf = shard_range(nbuf_total, rank, world)
    # threads for generating this rank's input: its share of the CPUs it may run on -- all of them divided by the ranks, or, once the rank
    # is bound to its GPU's NUMA node, that node's divided by the ranks that share the node (taken as evenly spread over the nodes)
    import glob
    nodes = max(1, len(glob.glob("/sys/devices/system/node/node[0-9]*")))
    sharing = world if getattr(args, "numa_node", None) is None else max(1, -(-world // nodes))
    ncpu = max(1, len(os.sched_getaffinity(0)) // sharing)
    iq_host, injected = synth.fill_range(first, nbuf, nthreads=ncpu)
    d_iq = torch.from_numpy(iq_host).cuda()
    torch.cuda.synchronize()
    sc = A.Scanner(local_rank)
    sc.set_outputs(A.OUT_PACKED)  # what travels to rank 0 is the packed form: record head + the GPU-decoded fields, 32 bytes
    compute = torch.cuda.current_stream()
    comm = torch.cuda.Stream()
    nbytes = d_iq.numel()
    on_device = dist.get_backend() == "nccl"
    # A rank that sits in one phase for longer than this prints rank, step and phase and ends with exit code 3: a stuck collective or
    # a lost peer must end the job as a failed process, never as a hang (and nothing here ever re-executes itself).
    wd = Watchdog(rank, float(os.environ.get("ADSB_AMD_WATCHDOG_S", "240")))

    # size the fixed gather buffers from a first scan (+25 % and the same on every rank)
    sc.submit(d_iq.data_ptr(), nbytes, BB, compute.cuda_stream, 0)
    n0 = len(sc.fetch_packed(0, copy=False))
    capt = torch.tensor([n0 + n0 // 4 + 4096], dtype=torch.int64, device="cuda" if on_device else "cpu")
    dist.all_reduce(capt, op=dist.ReduceOp.MAX)
    cap = int(capt.item())
    # how many ranks the collective backend itself has seen (a sum of ones over RCCL / gloo at set-up): the SCALE record shows it beside
    # ranks_seen, which counts the headers that arrived through the shared control page
    ones = torch.ones(1, dtype=torch.int64, device="cuda" if on_device else "cpu")
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    backend_ranks_seen = int(ones.item())
    # How the records reach rank 0's host.  Default: every GPU writes its records over its own PCIe link into a page-locked segment
    # of node-shared host memory and only 32-byte headers are gathered (shard.NodeGather) -- with a gather of the records themselves
    # (shard.RootGather: xGMI to rank 0's GPU, then its one host link) rank 0's PCIe link carries N x 9 MB per step and bounds the job
    # from N = 2 on.  --rccl-gather, or a node where the shared segment cannot be set up, uses the record gather.
    ng = None
    # the GPUs write the shared segments themselves: always over RCCL, and in the one-GPU rehearsal too (gloo only carries the set-up
    # collectives there; the record path -- registration of each rank's pages, N contexts copying into them, headers, credits -- is the real one)
    dma = on_device or bool(args.rehearse_on_one_gpu)
    if not args.rccl_gather:
        try:
            ng = NodeGather(cap, dtype=A.PACKED_DTYPE, device_writes=dma)
        except OSError as e:
            if rank == 0:
                print("bench: node-shared record segments unavailable (%s), gathering the records over RCCL" % e, file=sys.stderr)
    rg = RootGather(cap, dtype=A.PACKED_DTYPE)
    step_no = [0]
    # which transport every rank ended on (they fall back together by construction; the line shows it rank by rank, and so does the line a
    # failed job leaves: main())
    mine = "NodeGather" if ng is not None else "RootGather"
    by_rank = [None] * world
    dist.all_gather_object(by_rank, mine)
    args.transport_by_rank = by_rank

    def deliver_rccl(slot):
        """Records of `slot` -> rank 0 (collective).  Device path: scanner -> send buffer (device to device, on the side stream) ->
        RCCL gather -> rank 0's page-locked host memory; the compute stream only waits for the first of those copies."""
        if on_device:
            with torch.cuda.stream(comm):
                n = sc.fetch_device(slot, rg.records_ptr(), cap, comm.cuda_stream, packed=True)
                ev = comm.record_event()
                compute.wait_event(ev)  # the slot's device array may be overwritten by the next scan once it has been copied
                return rg.gather(n, first)
        rec = sc.fetch_packed(slot, copy=False)
        rg.host_records_view()[:len(rec)] = rec
        return rg.gather(len(rec), first)

    def deliver_node(slot):
        """Records of `slot` -> this rank's segment of the node-shared page-locked memory (device to host over the rank's own link, on
        the side stream), the step's header through the shared control page once the copy's event is done; rank 0 gets one view per rank, in recording order.  The segment is the one of step - 2:
        acquire() waits until rank 0 has released that step."""
        step = step_no[0]
        step_no[0] += 1
        wd.phase("acquire segment", step)
        ng.acquire(step)
        wd.phase("hand-over", step)
        if dma:
            with torch.cuda.stream(comm):
                n = sc.fetch_device(slot, ng.records_ptr(step), cap, comm.cuda_stream, packed=True)
                ev = comm.record_event()
                # (No compute.wait_event(ev): the dense array this copy reads is next written by the ordering pass in front of the scan kernel
                # two submits on, and that submit comes after ng.flush() has waited for this event on the host -- run() below --, which is
                # when the header is written too.  The wait on the stream cost every scan a packet.)
                return ng.gather(step, n, first, wait=False, event=ev)
        rec = sc.fetch_packed(slot, copy=False)
        ng.host_records_view(step)[:len(rec)] = rec
        return ng.gather(step, len(rec), first, wait=False)

    deliver = deliver_node if ng is not None else deliver_rccl
    seen = {"min": world}
    nbuf_of = [shard_range(nbuf_total, r, world)[1] for r in range(world)]

    def settle(out, resolver=None, keep=False):
        """rank 0 takes delivery of a step (shared segments: waits for the step's headers; record gather: already complete), optionally
        runs the sequential host half over it -- the per-rank parts one after the other, straight out of the shared segments --, and
        releases the step's segments (keep: copies the parts first).  Returns what was delivered."""
        if ng is not None and rank != 0:
            ng.flush()  # this rank's header of the step before (its copy finished during the scan submitted since)
        if out is None:
            return None
        wd.phase("take delivery", getattr(out, "step", -1))
        got = out.result() if hasattr(out, "result") else out
        if hasattr(out, "result"):
            seen["min"] = min(seen["min"], ng.ranks_seen)
            if resolver is not None:
                wd.phase("resolve", out.step)
                for (part, _first), nb in zip(got, nbuf_of):
                    resolver.feed(part, BB // 2, nb, collect=False)
            if keep:
                got = [(part.view(np.uint8).copy().view(part.dtype), f) for part, f in got]  # (numpy copies a structured array field by field: 6 ms for 9 MB)
            ng.release(out.step)
        elif resolver is not None and got is not None:
            resolver.feed(got, BB // 2, nbuf_total, collect=False)
        return got

    def run(steps, resolver=None):
        k_ms = 0.0
        out = prev = None
        wd.phase("submit", 0)
        sc.submit(d_iq.data_ptr(), nbytes, BB, compute.cuda_stream, 0)
        for i in range(1, steps):
            # (Round 6: the slot is scanned into again while the copy of its last records, step i - 2, may still be on its way -- a scan writes the
            # slot's raw regions only; the dense array the copy reads is written by the ordering pass of THIS scan, in front of the scan kernel of
            # step i + 1, which is submitted after the flush of the next trip.  The count of step i - 1 reaches the host some tens of microseconds into
            # the kernel submitted here, so waiting for that copy first, as until round 5, would leave the GPU without a kernel.)
            sc.submit(d_iq.data_ptr(), nbytes, BB, compute.cuda_stream, i & 1)
            if ng is not None:
                ng.flush()  # the copy of step i - 2 is done (or is waited for): its header goes out
            settle(prev, resolver)  # rank 0 takes (and releases) the step before, whose hand-over ran beside a scan
            prev = deliver((i - 1) & 1)
            k_ms += sc.timing((i - 1) & 1)[0]
        settle(prev, resolver)
        out = settle(deliver((steps - 1) & 1), resolver, keep=True)
        return out, k_ms + sc.timing((steps - 1) & 1)[0]

    def barrier():
        wd.phase("barrier")
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    run(SETUP_STEPS if not args.rehearse_on_one_gpu else 2)
    if args.warmup > 0:
        run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    rec, k_ms = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    # The same K steps once more with the sequential host half inside the loop: rank 0 resolves step k - 1 (N per-rank parts, one
    # core) while the GPUs scan step k; the ranks wait for rank 0's releases.  This is the whole recorded-file job, end to end.
    res_e2e = A.Resolver() if rank == 0 else None
    barrier()
    t0 = time.perf_counter()
    run(args.steps, res_e2e)
    barrier()
    elapsed_e2e = time.perf_counter() - t0

    # serial breakdown of one step (untimed region): scan, then gather, nothing overlapped
    barrier()
    ts = time.perf_counter()
    sc.submit(d_iq.data_ptr(), nbytes, BB, compute.cuda_stream, 0)
    torch.cuda.synchronize()
    tg = time.perf_counter()
    rec = settle(deliver(0), keep=True)
    torch.cuda.synchronize()
    te = time.perf_counter()
    rccl_serial = te - tg
    transports_agree = None
    if ng is not None:
        # the record gather over RCCL, once, for comparison (serial: scan, then gather, nothing overlapped)
        barrier()
        sc.submit(d_iq.data_ptr(), nbytes, BB, compute.cuda_stream, 0)
        torch.cuda.synchronize()
        tr = time.perf_counter()
        rec_rccl = deliver_rccl(0)
        torch.cuda.synchronize()
        rccl_serial = time.perf_counter() - tr
        if rank == 0:
            rec = NodeGather.concatenate(rec)
            transports_agree = bool(len(rec) == len(rec_rccl) and rec.tobytes() == rec_rccl.tobytes())
    # the round-1 definition for comparison: independent shards, records to each rank's own host, no gather
    run1 = make_runner(args, sc, d_iq, BB, compute.cuda_stream)
    run1(40)
    barrier()
    t1 = time.perf_counter()
    run1(args.steps)
    barrier()
    indep = time.perf_counter() - t1
    sharded_over_plain = None
    if args.sharded_step_on_one_rank:
        # the two loops in alternation, best window of each: what the hand-over costs a step, with the part's clock drift out of it
        best_s = best_p = 1e9
        for _ in range(3):
            barrier()
            t1 = time.perf_counter()
            run(args.steps)
            barrier()
            best_s = min(best_s, time.perf_counter() - t1)
            t1 = time.perf_counter()
            run1(args.steps)
            barrier()
            best_p = min(best_p, time.perf_counter() - t1)
        sharded_over_plain = round(best_p / best_s, 4)

    dev = "cuda" if on_device else "cpu"
    wd.phase("final reductions")
    t = torch.tensor([elapsed, indep, tg - ts, te - tg, rccl_serial, elapsed_e2e], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, indep, scan_serial, gather_serial, rccl_serial, elapsed_e2e = [float(x) for x in t.tolist()]
    c = torch.tensor([injected, k_ms / args.steps * 1e3], dtype=torch.float64, device=dev)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    injected_all, kernel_us_sum = int(c[0].item()), float(c[1].item())
    mine = torch.tensor([k_ms / args.steps], dtype=torch.float64, device=dev)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(per_rank, mine)
    per_rank = [float(x.item()) for x in per_rank]
    out = None
    if rank == 0:
        samples_all = nbuf_total * BB // 2
        kernel_ms = kernel_us_sum / 1e3 / world  # mean over ranks of the scan kernel's launch duration
        nrec = len(rec)
        alg_bytes = 2.0 * (samples_all / world) + 32.0 * (nrec / world)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        res = A.Resolver()
        t1 = time.perf_counter()
        accepted, fr, ac = res.feed(rec, BB // 2, nbuf_total, collect=bool(args.dump_stream))
        resolve_s = time.perf_counter() - t1
        if args.dump_stream:
            np.savez(args.dump_stream, records=rec, frames=fr, aircraft=ac, nbuf_total=nbuf_total)
        out = {
            "metric": "Msamples/s demodulated (1090ES u8 IQ -> Mode S frame records)",
            "value": round(samples_all * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "setup_steps": SETUP_STEPS, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8 in / u16 integer (bit-exact)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: recorded-file case, %d MiB recording = " % (world * args.mib) + WORKLOAD_1090 % (args.mib, nbuf)
                                   + "; rank r scans buffers [r*B/N, (r+1)*B/N), sorted records of all ranks delivered to rank 0's host in "
                                     "recording order: %s (%s)"
                                   % ("every GPU writes its records over its own PCIe link into a page-locked segment of node-shared host memory, "
                                      "a 32-byte header per rank and step through the shared control page (no collective per step)" if ng is not None else
                                      "one fixed-size gather of the records per step, device to device, then rank 0's GPU -> its host",
                                      "RCCL over xGMI" if on_device else "gloo: REHEARSAL on one GPU, numbers meaningless"),
                       "bytes_per_gpu": nbytes, "buffers_per_gpu": nbuf, "buffers_total": nbuf_total,
                       "sharding": "contiguous buffer ranges, no halo, no collective on the sample path; per step one %s" %
                                   ("header per rank in node-shared memory, written by the rank's host thread (records go host-side through node-shared memory too)" if ng is not None else "record gather"),
                       "record_transport": "node-shared page-locked host segments" if ng is not None else "RCCL record gather", "pipelined": True},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic(nbytes), "kernel": "scan1090_kernel", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes": int(alg_bytes), "note": "per GPU, mean over ranks"},
            "records_per_step": nrec, "frames_injected": injected_all, "decoded_msgs_per_step": int(accepted),
            "serial_step_ms": {"scan_and_order": round(scan_serial * 1e3, 4), "records_to_rank0_host": round(gather_serial * 1e3, 4),
                               "records_to_rank0_host_by_rccl_record_gather": round(rccl_serial * 1e3, 4)},
            "resolve_ms_rank0": round(resolve_s * 1e3, 2),
            # the whole job: scan on N GPUs + hand-over + the sequential host half on rank 0 (one core, no listener) inside the loop
            "end_to_end_msamples_per_s": round(samples_all * args.steps / elapsed_e2e / 1e6, 1),
            "end_to_end_ms_per_step": round(elapsed_e2e / args.steps * 1e3, 4),
            "end_to_end_note": "value counts delivery of the sorted records to rank 0's host (per-rank segments in recording order, which the "
                               "resolver reads in place); end_to_end adds rank 0 resolving every delivered step (one core) in the same loop",
            "ranks_seen": int(seen["min"]) if ng is not None else world,
            "rccl_ranks_seen": backend_ranks_seen, "collective_backend": "RCCL (torch.distributed nccl)" if on_device else "gloo (rehearsal)",
            "kernel_ms_by_rank": {"min": round(min(per_rank), 4), "max": round(max(per_rank), 4), "slowest_rank": int(per_rank.index(max(per_rank))),
                                  "all": [round(x, 4) for x in per_rank]},
            "independent_shards_value": round(samples_all * args.steps / indep / 1e6, 1),
            "record_transports_agree": transports_agree,
            "record_transport_by_rank": by_rank,
            "sharded_over_plain": sharded_over_plain,
            "gather_bytes_per_step": int(nrec * 32),
        }
    if ng is not None:
        ng.close()
    sc.close()
    wd.stop()
    return out


def bench_uat978(args, rank, local_rank, world, dist, A, synth, torch):
    """BASELINE configs[4]: UAT 978 u8 IQ (the reference's UAT input is u8, not i16) -> frames.  A step is one
    process_buffer over the rank's whole device-resident stream: discriminator + sync search, sync re-check + slicing +
    Reed-Solomon per match, the scan loop's decisions, records and the bit map of frames taken to the host, the host's walk over
    them (no up-calls).  Streams are independent replicas."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    piece = 64 << 20
    npieces = max(1, (args.mib << 20) // piece)
    cfg = synth.default_cfg978()
    dev = torch.empty(npieces * piece, dtype=torch.uint8, device="cuda")
    nthreads = max(1, min(npieces, len(os.sched_getaffinity(0)) // max(1, world), 16))
    first = None
    with ThreadPoolExecutor(nthreads) as ex:  # the generator releases the GIL (ctypes)
        for k0 in range(0, npieces, nthreads):
            ks = list(range(k0, min(npieces, k0 + nthreads)))
            for k, h in zip(ks, ex.map(lambda k: synth.fill978(rank * npieces + k, piece, cfg), ks)):
                first = h if first is None else first
                dev[k * piece:(k + 1) * piece].copy_(torch.from_numpy(h))
    torch.cuda.synchronize()
    nsamples = dev.numel() // 2
    u = A.Uat978(local_rank)

    def run(steps):
        scan = demod = 0.0
        for _ in range(steps):
            u.process_device(dev.data_ptr(), nsamples, collect=False)
            tm = u.timing()
            scan += tm["scan_ms"]
            demod += tm["demod_ms"]
        return scan, demod

    frames, consumed = u.process_device(dev.data_ptr(), nsamples)  # sizes the scratch; frame count for the report
    run(UAT_SETUP_STEPS)  # untimed: the GPU sat idle while the host generated the input (as SETUP_STEPS for the headline workload)
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
    elapsed_serial = time.perf_counter() - t0
    matches = (u.timing()["candidates"] - matches0) // args.steps
    nframes = len(frames)

    def run_pipelined(steps):
        """three calls in flight: the GPU halves of steps k + 1 and k + 2 (worker threads, own streams and buffer sets) under the host's walk of step k"""
        ahead = u.max_in_flight() - 1
        for k in range(min(ahead, steps)):
            u.submit_device(dev.data_ptr(), nsamples)
        for k in range(steps):
            if k + ahead < steps:
                u.submit_device(dev.data_ptr(), nsamples)
            u.collect(collect=False)

    if args.serial:
        elapsed = elapsed_serial
    else:
        run_pipelined(max(2, args.warmup))
        windows = []
        for _ in range(3):  # three timed windows of `steps` steps, the median reported: four host threads are involved and one
            barrier()       # descheduled worker shows up as a 20-30 % slower window on a shared box
            t0 = time.perf_counter()
            run_pipelined(args.steps)
            barrier()
            windows.append(time.perf_counter() - t0)
        elapsed = sorted(windows)[1]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([nframes], dtype=torch.int64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        nframes_all = int(c.item())
    else:
        nframes_all = nframes
    out = None
    if rank == 0:
        scan_k, demod_k = scan_ms / args.steps, demod_ms / args.steps
        tm = u.timing()
        # the dominant kernel by time is reported in "roofline"; the other one beside it
        alg_bytes = 2.0 * nsamples
        scan_roof = {"bound": "hbm", "achieved": round(alg_bytes / (scan_k * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg_bytes / (scan_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": measured_traffic("uat978:%d" % dev.numel()),
                     "kernel": "uat_scan_iq_kernel", "kernel_ms": round(scan_k, 4), "algorithmic_bytes": int(alg_bytes)}
        out = {
            "metric": "Msamples/s demodulated (UAT 978 u8 IQ -> Reed-Solomon-corrected frames)",
            "value": round(nsamples * world * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "setup_steps": UAT_SETUP_STEPS, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 in / u16 phase, GF(256) (bit-exact vs oracle; parity unpinned: dump978 is un-vendored)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: %d MiB synthetic UAT 978 u8 IQ per GPU (CPFSK h=0.6, 2 samples/bit, seed 0x978AD5B), "
                                   "phase LUT + discriminator + 18-bit sync search + 36-bit sync re-check + slicing + RS(30,18)/RS(48,34)/"
                                   "6xRS(92,72) and the dump978 scan-loop rules (which frames the loop takes: successor function + pointer jumping) on the GPU, "
                                   "the host walks the frames taken; parity unpinned (dump978 is un-vendored)"
                                   % (npieces * 64),
                       "bytes_per_gpu": int(dev.numel()), "sharding": "replicas only: one independent stream per GPU, no collective"},
            "roofline": scan_roof,
            # the path as a whole against the same bound: the stream's bytes (read once by the scan kernel; the demodulation's tiles are a few
            # per cent of them again and mostly cache hits) over one step of the pipelined loop, and over one call with nothing overlapped
            "path_roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes": int(alg_bytes),
                              "achieved": round(alg_bytes / (elapsed / args.steps) / 1e9, 1), "frac": round(alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                              "achieved_serial": round(alg_bytes / (elapsed_serial / args.steps) / 1e9, 1),
                              "frac_serial": round(alg_bytes / (elapsed_serial / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                              "note": "whole path per GPU: scan + ordering + demodulation + decisions + the host's walk; the demodulation kernel is "
                                      "bound by the number of its waves in flight (dependent chains of table look-ups and scalar decisions), not by "
                                      "memory: profiles/r05_uat978_demod_parts.txt"},
            "frames_per_step": nframes_all, "demod_kernel_ms": round(demod_k, 4), "matches_per_step_rank0": int(matches),
            "pipelined": not args.serial, "ms_per_step_serial": round(elapsed_serial / args.steps * 1e3, 4),
            "timing": "serial: one window of `steps` calls; pipelined (ms_per_step, value): median of three such windows",
            "host_wall_ms_last_step": tm["host_wall_ms"],
        }
        if demod_k > scan_k:
            # per-match kernel, instruction-bound (PMC, profiles/r05_uat978_rocprof_summary.txt): its algorithmic bytes are the phases of
            # the matches' frames, far below any HBM bound
            out["dominant_kernel"] = {"kernel": "uat_demod_kernel", "kernel_ms": round(demod_k, 4), "matches": int(matches),
                                      "us_per_1000_matches": round(demod_k * 1e3 / max(1, matches) * 1e3, 2),
                                      "note": "dominant by time; instruction-bound (sync re-check, slicing, Reed-Solomon per match; counters in "
                                              "profiles/r05_uat978_rocprof_summary.txt), not bandwidth-bound: the HBM roofline "
                                              "above is the scan kernel's, the only kernel of this path that streams the input"}
        if world == 1 and args.cpu_buffers > 0:
            from oracle import oracle_py as O
            phi = O.phase_lut978()[first.view(np.uint16)]
            t1 = time.perf_counter()
            want, _ = O.process_buffer978(phi)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(first.size / 2 / dt / 1e6, 1), "unit": "Msamples/s", "cores": 1, "kind": "port",
                                   "sample": "oracle978 process_buffer over the first 64 MiB of the same stream (%.2f s), LUT map excluded; %d frames"
                                             % (dt, len(want))}
    u.close()
    del dev
    torch.cuda.empty_cache()
    return out


def bench_uat978_one_stream(args, rank, local_rank, world, dist, A, synth, torch):
    """SURVEY section 8e, "shard with a halo": ONE process_buffer over a stream of world x args.mib, cut over the ranks
    (adsb_amd_uat_part_scan / _finish; libadsb_amd/shard.py: uat_part).  Every rank searches and demodulates its part (with 64 samples
    before it and three maximum frames behind it) independently; which frames the scan loop takes in a part depends on where the loop
    stands when it comes in, so one 8-byte number travels from rank to rank (send / recv) before a rank decides and makes its up-calls.
    A step is that whole job once; the steps are not overlapped with each other here (the chain of decisions is serial: about 0.2 ms a rank)."""
    import numpy as np
    from libadsb_amd import shard
    from libadsb_amd.shard import Watchdog
    piece = 64 << 20
    npieces = max(1, (args.mib << 20) // piece)
    cfg = synth.default_cfg978()
    own_bytes = npieces * piece
    n_total = world * own_bytes // 2
    w0, w1, b, e, last = shard.uat_part(n_total, rank, world)
    wd = Watchdog(rank, float(os.environ.get("ADSB_AMD_WATCHDOG_S", "240")))
    on_device = dist.get_backend() == "nccl"
    buf = torch.empty(2 * (w1 - w0), dtype=torch.uint8, device="cuda")
    lead, tail = b - w0, w1 - e
    if lead:
        buf[:2 * lead].copy_(torch.from_numpy(synth.fill978(rank * npieces - 1, piece, cfg)[-2 * lead:].copy()))
    for k in range(npieces):
        buf[2 * lead + k * piece:2 * lead + (k + 1) * piece].copy_(torch.from_numpy(synth.fill978(rank * npieces + k, piece, cfg)))
    if tail:
        buf[2 * lead + own_bytes:].copy_(torch.from_numpy(synth.fill978((rank + 1) * npieces, piece, cfg)[:2 * tail].copy()))
    torch.cuda.synchronize()
    u = A.Uat978(local_rank)
    word = torch.zeros(1, dtype=torch.int64, device="cuda" if on_device else "cpu")

    def step(collect=False):
        frames, consumed, _ = shard.uat_chain_step(u, dist, word, rank, world, (w0, w1, b, e, last), buf.data_ptr(), collect=collect, phase=wd.phase)
        return frames, consumed

    def barrier():
        wd.phase("barrier")
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    frames, consumed = step(collect=True)  # untimed: sizes the scratch, counts the frames (a Python up-call each)
    for _ in range(max(1, min(args.warmup, 3))):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    dev = "cuda" if on_device else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([len(frames), consumed if last else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    out = None
    if rank == 0:
        elapsed = float(t.item())
        out = {"value": round(n_total * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "steps": args.steps, "samples_in_the_stream": int(n_total), "frames": int(c[0].item()), "consumed": int(c[1].item()),
               "note": "one process_buffer over the whole stream per step; every rank's search and demodulation in parallel, the decisions rank after rank"}
        if n_total < (1 << 30):
            # small enough for one call on this GPU: the same stream, whole
            whole = np.concatenate([synth.fill978(k, piece, cfg) for k in range(world * npieces)])
            wdev = torch.from_numpy(whole).cuda()
            ref = A.Uat978(local_rank)
            want_frames, want_consumed = ref.process_device(wdev.data_ptr(), whole.size // 2)
            ref.close()
            out["equals_one_call"] = bool(len(want_frames) == out["frames"] and want_consumed == out["consumed"])
    wd.stop()
    u.close()
    return out


if __name__ == "__main__":
    main()
