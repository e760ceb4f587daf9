"""In-process A/B of kernel variants selected by environment knobs (read per submit): interleaved rounds, median kernel ms.

    python tools/ab.py "lib=ab_ship/a.so" "lib=ab_ship/b.so" ...
    AB_RATE=24 python tools/ab.py ...      the 2.4 MS/s mode and its workload
"""
import os
import sys
import statistics

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import libadsb_amd as A  # noqa: E402
from libadsb_amd import synth  # noqa: E402

variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in sys.argv[1:]] or [{}]
# a variant may name its own build of the library: lib=ab_ship/foo.so (see tools/build_variant.sh)
BB = A.REF_BUFFER_BYTES
nbuf = 4096
RATE = int(os.environ.get("AB_RATE", "20"))
MODE = A.MODE_2400 if RATE == 24 else A.MODE_2000
iq, _ = synth.fill_range(0, nbuf, nthreads=16, rate_x10=RATE)
d = torch.from_numpy(iq).cuda()
torch.cuda.synchronize()
import ctypes as C
scanners = {}
def scanner_for(v):
    path = v.get("lib")
    if path not in scanners:
        if path:
            A._lib = None
            A.LIB_PATH = os.path.abspath(path)
        scanners[path] = A.Scanner(0, mode=MODE)
    return scanners[path]
st = torch.cuda.current_stream().cuda_stream
keys = sorted({k for v in variants for k in v})
res = [[] for _ in variants]
nrec = [0] * len(variants)
for rnd in range(int(os.environ.get("AB_ROUNDS", "7"))):
    for i, v in enumerate(variants):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update({k: x for k, x in v.items() if k != "lib"})
        sc = scanner_for(v)
        for _ in range(3):
            sc.submit(d.data_ptr(), d.numel(), BB, st, 0)
            r = sc.fetch(0, copy=False)
            nrec[i] = len(r)
            if rnd:
                res[i].append(sc.timing(0)[0])
for v, r, n in zip(variants, res, nrec):
    print("%-40s kernel_ms median %.4f min %.4f  (records %d)" % (v, statistics.median(r), min(r), n))
