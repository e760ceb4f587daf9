"""What the scan kernels cost away from the BASELINE workload: kernel ms per GiB at noise +-3 / 10 / 20 / 40 x mean frame spacing 300 / 2000 /
100000 samples, both rates, with the records of the first 64 buffers compared with the oracle at every point (profiles/r04_sensitivity.txt).
    python tools/sensitivity.py"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libadsb_amd as A
from libadsb_amd import synth
from oracle import oracle_py as O
BB = A.REF_BUFFER_BYTES
NBUF, NCHECK = 4096, 64
st = torch.cuda.current_stream().cuda_stream
ncpu = max(1, len(os.sched_getaffinity(0)))


def kernel_ms(sc, d, secs=0.5):
    t0 = time.perf_counter(); ks = []; i = 1
    sc.submit(d.data_ptr(), d.numel(), BB, st, 0)
    while time.perf_counter() - t0 < secs:
        sc.submit(d.data_ptr(), d.numel(), BB, st, i & 1)
        rec = sc.fetch_packed((i - 1) & 1, copy=False); ks.append(sc.timing((i - 1) & 1)[0]); i += 1
    rec = sc.fetch_packed((i - 1) & 1, copy=False)
    return statistics.median(ks[len(ks) // 2:]), len(rec), (time.perf_counter() - t0) / i * 1e3


print("%-6s %-8s %-5s %10s %9s %10s %11s %11s %8s" % ("noise", "spacing", "rate", "kernel ms", "% of HBM", "step ms", "records", "per chunk", "oracle"))
for noise in (3, 10, 20, 40):
    for spacing in (300, 2000, 100000):
        cfg = synth.default_cfg(noise_amp=noise, mean_spacing=spacing)
        for rate in (20, 24):
            iq, inj = synth.fill_range(0, NBUF, nthreads=ncpu, rate_x10=rate, cfg=cfg)
            d = torch.from_numpy(iq).cuda(); torch.cuda.synchronize()
            sc = A.Scanner(0, mode=rate); sc.set_outputs(A.OUT_PACKED)
            k, nrec, step = kernel_ms(sc, d)
            sc.set_outputs(A.OUT_RECORDS | A.OUT_DECODED)
            got = sc.scan(iq[:NCHECK * BB], BB)
            if rate == 20:
                want = O.expected_records(iq[:NCHECK * BB], BB, dtype=A.RECORD_DTYPE)
                o = O.Oracle1090()
                for b in range(NCHECK):
                    c = iq[b * BB:(b + 1) * BB]
                    O.lib().oracle1090_handle_data(o._h, c.ctypes.data, c.size, None, None)
                s = o.stats()
                extra = "  reference loop per 4096 positions: stage-1 survivors %.1f, stage-2 candidates %.2f, accepted %.2f" % (
                    s["stage1_pass"] / (NCHECK * 32.0), s["stage2_pass"] / (NCHECK * 32.0), s["accepted"] / (NCHECK * 32.0))
            else:
                want = O.expected_records2400(iq[:NCHECK * BB], BB, dtype=A.RECORD_DTYPE, nthreads=min(16, ncpu))
                extra = ""
            same = len(got) == len(want) and got.tobytes() == want.tobytes()
            alg = 2.0 * (NBUF * BB // 2) + 32.0 * nrec
            print("%-6d %-8d %-5s %10.4f %9.1f %10.4f %11d %11.2f %8s%s" % (noise, spacing, "2.0" if rate == 20 else "2.4", k, alg / (k * 1e-3) / 8e12 * 100, step, nrec,
                                                                         nrec / (NBUF * 32.0), "equal" if same else "DIFFERS", extra), flush=True)
            sc.close(); del d
            torch.cuda.empty_cache()
