"""Kernel ms of build variants at one generator setting (sustained, like tools/sustained_ab.py).
    AB_RATE=24 AB_NOISE=3 AB_SPACING=300 python tools/ab_setting.py ab_ship/a.so ab_ship/b.so"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libadsb_amd as A
from libadsb_amd import synth
BB = A.REF_BUFFER_BYTES
RATE = int(os.environ.get("AB_RATE", "20"))
cfg = synth.default_cfg(noise_amp=int(os.environ.get("AB_NOISE", "3")), mean_spacing=int(os.environ.get("AB_SPACING", "2000")))
iq, _ = synth.fill_range(0, 4096, nthreads=16, rate_x10=RATE, cfg=cfg)
d = torch.from_numpy(iq).cuda(); torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
for rnd in range(2):
    for path in sys.argv[1:]:
        A._lib = None; A.LIB_PATH = os.path.abspath(path)
        sc = A.Scanner(0, mode=RATE); sc.set_outputs(A.OUT_PACKED)
        t0 = time.perf_counter(); ks = []; i = 1
        sc.submit(d.data_ptr(), d.numel(), BB, st, 0)
        while time.perf_counter() - t0 < 1.0:
            sc.submit(d.data_ptr(), d.numel(), BB, st, i & 1)
            n = len(sc.fetch_packed((i - 1) & 1, copy=False)); ks.append(sc.timing((i - 1) & 1)[0]); i += 1
        sc.fetch_packed((i - 1) & 1, copy=False)
        print("%-22s round %d: kernel ms %.4f  records %d" % (os.path.basename(path), rnd, statistics.median(ks[len(ks) // 2:]), n), flush=True)
        sc.close()
