"""Recorded-file replay rate of the batch path (adsb_amd_handler_replay_file) from the page cache, 1 GiB, a few repeats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libadsb_amd as A
from libadsb_amd import synth
iq, _ = synth.fill_range(0, 4096, nthreads=16)
path = "/dev/shm/replay_rate.dat"
iq.tofile(path)
for k in range(5):
    h = A.Handler1090(); t = time.perf_counter(); n, _, _ = h.replay_file(path, collect=False); dt = time.perf_counter() - t; h.close()
    print("run %d: %d frames, %.1f ms = %.1f GiB/s" % (k, n, dt * 1e3, 1.0 / dt))
os.unlink(path)
