"""kernel ms of library builds on three inputs (BASELINE, noise +-20, spacing 300), several contexts each, means.
    python tools/ab_dense.py ab_ship/a.so ab_ship/b.so"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libadsb_amd as A
from libadsb_amd import synth
BB = A.REF_BUFFER_BYTES
st = torch.cuda.current_stream().cuda_stream
RATE = int(os.environ.get("AB_RATE", "20"))
inputs = {}
for name, over in (("baseline", {}), ("noise20", {"noise_amp": 20}), ("spacing300", {"mean_spacing": 300})):
    iq, _ = synth.fill_range(0, 4096, nthreads=16, rate_x10=RATE, cfg=synth.default_cfg(**over))
    inputs[name] = torch.from_numpy(iq).cuda()
torch.cuda.synchronize()
res = {}
for rnd in range(int(os.environ.get("AB_CONTEXTS", "3"))):
    for path in sys.argv[1:]:
        A._lib = None; A.LIB_PATH = os.path.abspath(path)
        sc = A.Scanner(0, mode=RATE); sc.set_outputs(A.OUT_PACKED)
        for name, d in inputs.items():
            t0 = time.perf_counter(); ks = []; i = 1
            sc.submit(d.data_ptr(), d.numel(), BB, st, 0)
            while time.perf_counter() - t0 < 0.5:
                sc.submit(d.data_ptr(), d.numel(), BB, st, i & 1)
                sc.fetch_packed((i - 1) & 1, copy=False); ks.append(sc.timing((i - 1) & 1)[0]); i += 1
            n = len(sc.fetch_packed((i - 1) & 1, copy=False))
            res.setdefault((os.path.basename(path), name), []).append(statistics.median(ks[len(ks) // 2:]))
        sc.close()
for (lib, name), v in sorted(res.items()):
    print("%-14s %-11s kernel ms mean %.4f  (%s)" % (lib, name, sum(v) / len(v), " ".join("%.4f" % x for x in v)))
