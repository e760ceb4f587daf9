"""Latency of one live-sized UAT HandleData call (262 144 B of host IQ, PCIe inclusive): python tools/uat_live_latency.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libadsb_amd as A  # noqa: E402
from libadsb_amd import synth  # noqa: E402

iq = synth.fill978(0, 64 * 262144, synth.default_cfg978(mean_gap_bits=20000))
u = A.Uat978()
L, h = A.lib(), u._h
t = []
nframes = 0
for k in range(64):
    part = iq[k * 262144:(k + 1) * 262144]
    t0 = time.perf_counter()
    rc = L.adsb_amd_uat_handle_data(h, part.ctypes.data, part.size, None, None)
    t.append(time.perf_counter() - t0)
    assert rc == 0
t = sorted(t[4:])
print("UAT HandleData 262144 B: median %.3f ms, p90 %.3f ms, min %.3f ms (budget at 2.083 MS/s: 62.9 ms)" % (t[len(t) // 2] * 1e3, t[int(len(t) * 0.9)] * 1e3, t[0] * 1e3))
