"""Soak test: many HandleData calls of varying sizes through both handlers; device and host memory must stay flat.
    python tools/soak.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import psutil  # noqa: E402
import libadsb_amd as A  # noqa: E402
from libadsb_amd import synth  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
BB = A.REF_BUFFER_BYTES
iq1090, _ = synth.fill_range(0, 64, nthreads=8)
iq978 = synth.fill978(0, 64 * BB, synth.default_cfg978())
h = A.Handler1090(0)
u = A.Uat978(0)
# recorded-file replays with a handler of their own now and then (the replay's staging is the handler's: made by the first pass, released with it)
replay_path = "/dev/shm/adsb_amd_soak_%d.test.dat" % os.getpid()
np.concatenate([iq1090, iq1090, iq1090, iq1090[:8 * BB]]).tofile(replay_path)  # 200 buffers: four batches, the last one partial
replays = 0
L = A.lib()
proc = psutil.Process()
rng = np.random.default_rng(1)
t0 = time.time()
calls = frames = 0
report = t0
base = None
while time.time() - t0 < secs:
    k = int(rng.integers(0, 60))
    nb = int(rng.choice([1, 1, 1, 2, 4]))
    part = iq1090[k * BB:(k + nb) * BB]
    n = L.adsb_amd_handler_handle_data(h._h, part.ctypes.data, part.size, BB, None, None)
    assert n >= 0
    frames += n
    part = iq978[k * BB:(k + nb) * BB]
    if rng.random() < 0.1:
        part = part[:int(rng.integers(2, part.size)) & ~1]
    assert L.adsb_amd_uat_handle_data(u._h, part.ctypes.data, part.size, None, None) == 0
    calls += 2
    if calls % 80 == 0:
        hr = A.Handler1090(0)
        n1, _, _ = hr.replay_file(replay_path, collect=False)
        n2, _, _ = hr.replay_file(replay_path, first_buffer=int(rng.integers(0, 150)), collect=False)
        assert n1 > 0 and n2 > 0
        hr.close()
        replays += 2
    if time.time() - report > 10:
        free, total = torch.cuda.mem_get_info()
        rss = proc.memory_info().rss
        if base is None:
            base = (free, rss)
        print("t=%3.0fs calls=%d frames=%d  device used=%.1f MiB (delta %+.1f)  host rss=%.1f MiB (delta %+.1f)"
              % (time.time() - t0, calls, frames, (total - free) / 2**20, (base[0] - free) / 2**20, rss / 2**20, (rss - base[1]) / 2**20), flush=True)
        report = time.time()
os.unlink(replay_path)
free, total = torch.cuda.mem_get_info()
rss = proc.memory_info().rss
print("replays of a 200-buffer file with handlers of their own: %d" % replays)
print("done: %d calls; device delta %+.1f MiB, host rss delta %+.1f MiB since the first report" % (calls, (base[0] - free) / 2**20, (rss - base[1]) / 2**20))
assert abs(base[0] - free) < 64 * 2**20 and rss - base[1] < 64 * 2**20, "memory is growing"
