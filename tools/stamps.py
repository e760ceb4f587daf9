"""Time line of the pipelined scan loop without a profiler: a measurement build (-DADSB_AMD_DIAG_BUILD=1 -DDIAG_STAMPS=1, diag.hip.h) lets every kernel note when its first wave came in
and its last went out (100 MHz clock); this prints the steady-state averages of scan, gap, ordering pass, gap.
    EXTRA_FLAGS="-DADSB_AMD_DIAG_BUILD=1 -DDIAG_STAMPS=1" tools/build_variant.sh WORK stamps && python tools/stamps.py ab_ship/stamps.so [timing_every]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libadsb_amd as A
from libadsb_amd import synth
A.LIB_PATH = os.path.abspath(sys.argv[1])
every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
BB = A.REF_BUFFER_BYTES
RATE = int(os.environ.get('AB_RATE', '20'))
iq, _ = synth.fill_range(0, 4096, nthreads=16, rate_x10=RATE)
d = torch.from_numpy(iq).cuda(); torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
sc = A.Scanner(0, mode=RATE); sc.set_outputs(A.OUT_PACKED); sc.set_timing(every)
N = 128 * 6  # the library keeps the last 128 launches
t0 = time.perf_counter()
sc.submit(d.data_ptr(), d.numel(), BB, st, 0)
for i in range(1, N):
    sc.submit(d.data_ptr(), d.numel(), BB, st, i & 1)
    sc.fetch_packed((i - 1) & 1, copy=False)
sc.fetch_packed((N - 1) & 1, copy=False)
el = time.perf_counter() - t0
buf = (C.c_ulonglong * (128 * 4))(); n = C.c_uint()
sc._l.adsb_amd_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]
assert sc._l.adsb_amd_debug_stamps(sc._h, buf, C.byref(n)) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
assert n.value % 128 == 0  # so that the ring is in launch order
us = lambda x: float(np.mean(x)) / 100.0
print("%d launches, timing events on every %s: step %.1f us by the host's clock" % (n.value, every or "none", el / N * 1e6))
print("scan first wave in -> last wave out %.1f us | -> ordering pass in %.1f us | ordering pass %.1f us | -> next scan in %.1f us | sum %.1f us" % (
    us(s[:, 1] - s[:, 0]), us(s[:, 2] - s[:, 1]), us(s[:, 3] - s[:, 2]), us(s[1:, 0] - s[:-1, 3]), us(s[1:, 0] - s[:-1, 0])))
