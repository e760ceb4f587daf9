"""PCIe-inclusive rate of the synchronous host-buffer entry points (what HandleData pays): for DESIGN.md, not bench.py's value."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import libadsb_amd as A
from libadsb_amd import synth
BB = A.REF_BUFFER_BYTES
for nbuf in (1, 16, 4096):
    iq, _ = synth.fill_range(0, nbuf, nthreads=16)
    sc = A.Scanner(0)
    sc.scan(iq, BB)
    t = time.perf_counter(); reps = 20 if nbuf < 100 else 3
    for _ in range(reps): r = sc.scan(iq, BB)
    dt = (time.perf_counter() - t) / reps
    h = A.Handler1090(0)
    h.handle_data(iq[:BB]);
    t = time.perf_counter()
    n = h._l.adsb_amd_handler_handle_data(h._h, iq.ctypes.data, iq.size, BB, None, None)
    dth = time.perf_counter() - t
    print("buffers=%d  scan(host in, records out): %.3f ms = %.1f Msamples/s | handle_data incl. host resolve: %.3f ms = %.1f Msamples/s (%d accepted)"
          % (nbuf, dt * 1e3, nbuf * BB / 2 / dt / 1e6, dth * 1e3, nbuf * BB / 2 / dth / 1e6, n))
    sc.close(); h.close()

# live-sized calls out of page-locked ring slots (adsb_amd_host_alloc) vs ordinary memory
iq, _ = synth.fill_range(0, 8, nthreads=8)
pin = A.PinnedBuffer(8 * BB)
pin.array[:] = iq
for name, src in (("pageable", iq), ("page-locked", pin.array)):
    h = A.Handler1090(0)
    ts = []
    for rep in range(6):
        for k in range(8):
            part = src[k * BB:(k + 1) * BB]
            t = time.perf_counter()
            h._l.adsb_amd_handler_handle_data(h._h, part.ctypes.data, part.size, BB, None, None)
            ts.append(time.perf_counter() - t)
    ts = sorted(ts[8:])
    print("HandleData 262144 B from %s memory: median %.3f ms, min %.3f ms" % (name, ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
    h.close()
pin.close()
