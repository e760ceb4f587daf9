"""When do the waves of one scan launch finish?  (measurement build: -DADSB_AMD_DIAG_BUILD=1 -DDIAG_STAMPS=1, diag.hip.h)  Per XCD (workgroup index % 8) and per work counter: the time of the
last wave out, relative to the first wave in.
    python tools/stamps_waves.py ab_ship/stamps.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libadsb_amd as A
from libadsb_amd import synth
A.LIB_PATH = os.path.abspath(sys.argv[1])
BB = A.REF_BUFFER_BYTES
iq, _ = synth.fill_range(0, 4096, nthreads=16)
d = torch.from_numpy(iq).cuda(); torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
sc = A.Scanner(0); sc.set_outputs(A.OUT_PACKED); sc.set_timing(0)
N = 128 * 4
sc.submit(d.data_ptr(), d.numel(), BB, st, 0)
for i in range(1, N):
    sc.submit(d.data_ptr(), d.numel(), BB, st, i & 1)
    sc.fetch_packed((i - 1) & 1, copy=False)
sc.fetch_packed((N - 1) & 1, copy=False)
G = 8192
buf = (C.c_ulonglong * (4 * G))()
sc._l.adsb_amd_debug_stamps_raw.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
for launch in (100, 101, 102, 103, 104, 105):
    assert sc._l.adsb_amd_debug_stamps_raw(sc._h, launch, buf) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(4, G).astype(np.int64)
    t_in, t_out = s[0][:4096], s[1][:4096]
    ok = t_in > 0
    t0 = t_in[ok].min()
    print("launch %d: first wave in 0, last wave in %.1f us; waves out: first %.1f us, median %.1f, last %.1f" % (
        launch, (t_in[ok].max() - t0) / 100.0, (t_out[ok].min() - t0) / 100.0, float(np.median(t_out[ok]) - t0) / 100.0, (t_out[ok].max() - t0) / 100.0))
    w = np.arange(4096)
    print("   waves out, percentiles 0 1 5 25 50 75 95 99 100: " + " ".join("%.1f" % ((np.percentile(t_out[ok], q) - t0) / 100.0) for q in (0, 1, 5, 25, 50, 75, 95, 99, 100)))
    for x in range(8):
        for k in range(4):
            m = ok & (w % 8 == x) & ((w // 8) % 4 == k)
            if x < 0: print("   XCD slot %d counter %d: waves out min %.1f median %.1f max %.1f" % (x, k, (t_out[m].min() - t0) / 100.0, float(np.median(t_out[m]) - t0) / 100.0, (t_out[m].max() - t0) / 100.0))
    for x in range(8):
        m = ok & (w % 8 == x)
        print("   XCD slot %d: waves out median %.1f last %.1f us, by counter %s" % (x, float(np.median(t_out[m]) - t0) / 100.0, (t_out[m].max() - t0) / 100.0,
              " ".join("%.1f" % ((t_out[m & ((w // 8) % 4 == k)].max() - t0) / 100.0) for k in range(4))))
