"""What the UAT kernels of one call take (the handler's own events) when calls run back to back and when three are in flight.
    python tools/uat_pipelined_times.py [MiB] [steps]
If a kernel's span is the same in both cases it ran alone on the chip in the pipelined case too (its neighbours waited); if it is longer it
shared the chip.  The step is printed beside the sum of the spans.
"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import libadsb_amd as A  # noqa: E402
from libadsb_amd import synth  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
piece = 64 << 20
npieces = max(1, (mib << 20) // piece)
cfg = synth.default_cfg978()
dev = torch.empty(npieces * piece, dtype=torch.uint8, device="cuda")
with ThreadPoolExecutor(min(16, npieces)) as ex:
    for k, h in enumerate(ex.map(lambda k: synth.fill978(k, piece, cfg), range(npieces))):
        dev[k * piece:(k + 1) * piece].copy_(torch.from_numpy(h))
torch.cuda.synchronize()
n = dev.numel() // 2
u = A.Uat978(0)
u.process_device(dev.data_ptr(), n)
keys = ("scan_ms", "demod_ms")


def serial(k):
    acc = dict.fromkeys(keys, 0.0)
    dec = 0.0
    t0 = time.perf_counter()
    for _ in range(k):
        u.process_device(dev.data_ptr(), n, collect=False)
        tm = u.timing()
        for x in keys:
            acc[x] += tm[x]
        dec += tm["host_wall_ms"]["decide_kernels"]
    return (time.perf_counter() - t0) / k * 1e3, {x: acc[x] / k for x in keys}, dec / k


def pipelined(k):
    acc = dict.fromkeys(keys, 0.0)
    dec = 0.0
    ahead = u.max_in_flight() - 1
    t0 = time.perf_counter()
    for i in range(min(ahead, k)):
        u.submit_device(dev.data_ptr(), n)
    for i in range(k):
        if i + ahead < k:
            u.submit_device(dev.data_ptr(), n)
        u.collect(collect=False)
        tm = u.timing()
        for x in keys:
            acc[x] += tm[x]
        dec += tm["host_wall_ms"]["decide_kernels"]
    return (time.perf_counter() - t0) / k * 1e3, {x: acc[x] / k for x in keys}, dec / k


serial(20), pipelined(20)
for name, f in (("back to back", serial), ("three in flight", pipelined), ("back to back", serial), ("three in flight", pipelined)):
    step, t, dec = f(steps)
    print("%-16s step %.4f ms | scan span %.4f  demod span %.4f  decision kernels' span %.4f | spans together %.4f" % (
        name, step, t["scan_ms"], t["demod_ms"], dec, t["scan_ms"] + t["demod_ms"] + dec), flush=True)
