"""What the host half costs when it is handed ONLY the records it accepts (the most a GPU-side pre-resolution of the skip-ahead could leave it with),
against what it costs on all records of a step.  1 GiB of the bench workload; the records come from a GPU scan (packed form).
    python tools/resolver_floor.py            -> profiles/r06_resolver.txt (run on the GPU box: its host is where the resolver's figure is quoted)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import libadsb_amd as A
from libadsb_amd import synth
BB = A.REF_BUFFER_BYTES
nbuf = 4096
iq, _ = synth.fill_range(0, nbuf, nthreads=16)
d = torch.from_numpy(iq).cuda()
sc = A.Scanner(0)
sc.set_outputs(A.OUT_PACKED)
sc.submit(d.data_ptr(), d.numel(), BB, torch.cuda.current_stream().cuda_stream, 0)
rec = sc.fetch_packed(0, copy=True)
sc.close()


def best_of(records, reps=6):
    r = A.Resolver()
    n = r.feed(records, BB // 2, nbuf, collect=False)[0]  # (the first stretch makes the helper thread and fills the tables)
    t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        r.feed(records, BB // 2, nbuf, collect=False)
        t = min(t, time.perf_counter() - t0)
    r.close()
    return n, t * 1e3


n_all, ms_all = best_of(rec)
# which records the resolver accepts: one collecting pass (a Python up-call per frame: slow, once)
r = A.Resolver()
n, fr, _ = r.feed(rec, BB // 2, nbuf, collect=True)
r.close()
acc_key = fr["offset"].astype(np.uint64) * 2 + (fr["pass"].astype(np.uint64) == 2)
rec_key = (rec["buffer"].astype(np.uint64) * (BB // 2) + rec["offset"].astype(np.uint64)) * 2 + (rec["flags"] & 1).astype(np.uint64)
keep = np.isin(rec_key, acc_key)
sub = np.ascontiguousarray(rec[keep])
n_sub, ms_sub = best_of(sub)
stateless = np.ascontiguousarray(rec[(rec["flags"] & A.F_NEEDS_ICAO) == 0])
n_st, ms_st = best_of(stateless)
print("records of one step (1 GiB, bench workload): %d, of which accepted %d (%.1f %%); AP-type (ICAO-gated) records %d" % (len(rec), n_all, 100.0 * n_all / len(rec), int(((rec["flags"] & A.F_NEEDS_ICAO) != 0).sum())))
print("host half, all records                          : %.3f ms  (%.2f ns per record, %.2f ns per accepted frame)" % (ms_all, ms_all * 1e6 / len(rec), ms_all * 1e6 / n_all))
print("host half, ONLY the records it accepts          : %.3f ms  (%d records, %d accepted: %.2f ns per accepted frame)" % (ms_sub, len(sub), n_sub, ms_sub * 1e6 / max(1, n_sub)))
print("host half, only DF11/17 records (no ICAO gating): %.3f ms  (%d records, %d accepted)" % (ms_st, len(stateless), n_st))
