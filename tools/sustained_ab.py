"""Kernel ms of build variants in SUSTAINED operation (what bench.py measures): every variant runs its scans back to back over two
slots for `seconds`, the median of the second half of the run is reported; variants one after the other with a pause.
    python tools/sustained_ab.py ab_ship/a.so ab_ship/b.so ...     (AB_SECONDS=2)"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libadsb_amd as A
from libadsb_amd import synth
BB = A.REF_BUFFER_BYTES
secs = float(os.environ.get("AB_SECONDS", "2"))
RATE = int(os.environ.get('AB_RATE', '20'))
iq, _ = synth.fill_range(0, 4096, nthreads=16, rate_x10=RATE)
d = torch.from_numpy(iq).cuda(); torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
scanners = []
PACKED = os.environ.get("AB_PACKED", "1") == "1"  # what bench.py moves: the packed form
def timing(sc, slot):
    return sc.timing_if_timed(slot) if hasattr(sc, "timing_if_timed") else sc.timing(slot)
def fetch(sc, slot):
    return sc.fetch_packed(slot, copy=False) if PACKED else sc.fetch_decoded(slot, copy=False)
for path in sys.argv[1:]:
    A._lib = None; A.LIB_PATH = os.path.abspath(path)
    sc = A.Scanner(0, mode=RATE)
    if PACKED: sc.set_outputs(A.OUT_PACKED)
    if "AB_TIMING_EVERY" in os.environ and hasattr(sc._l, "adsb_amd_set_timing"): sc.set_timing(int(os.environ["AB_TIMING_EVERY"]))
    scanners.append((path, sc))
for rnd in range(2):
    for path, sc in scanners:
        t0 = time.perf_counter(); ks = []; ts = []; tf = []; tsub = []; by_slot = ([], [])
        sc.submit(d.data_ptr(), d.numel(), BB, st, 0); i = 1
        while time.perf_counter() - t0 < secs:
            ta = time.perf_counter()
            sc.submit(d.data_ptr(), d.numel(), BB, st, i & 1)
            tb = time.perf_counter()
            fetch(sc, (i - 1) & 1); tc = time.perf_counter(); tsub.append(tb - ta); tf.append(tc - tb); tm = timing(sc, (i - 1) & 1); i += 1
            if tm: ks.append(tm[0]); ts.append(tm[1]); by_slot[i & 1].append(tm[0])
        fetch(sc, (i - 1) & 1)
        el = time.perf_counter() - t0
        print("%-22s round %d: kernel ms median (second half) %.4f  scan start -> count on the host %.4f  step %.4f ms" % (os.path.basename(path), rnd, statistics.median(ks[len(ks) // 2:]), statistics.median(ts[len(ts) // 2:]), el / i * 1e3) + "  slots %.4f %.4f" % tuple(statistics.median(b[len(b) // 2:]) if b else 0 for b in by_slot) + "  (host: submit %.1f us, fetch %.1f us)" % (statistics.median(tsub) * 1e6, statistics.median(tf) * 1e6), flush=True)
        time.sleep(0.3)
