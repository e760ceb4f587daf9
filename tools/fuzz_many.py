"""One-off wide fuzz: the fuzz tests of the suite with many more seeds.  python tools/fuzz_many.py [n1090] [n978]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libadsb_amd as A  # noqa: E402
import test_gpu_parity as T1  # noqa: E402
import test_uat978_gpu as T2  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sc = A.Scanner()
t = time.time()
for seed in range(100, 100 + n1):
    T1.test_fuzzed_threshold_cases(sc, seed)
print("1090: %d fuzz seeds identical to the oracle (%.0f s)" % (n1, time.time() - t), flush=True)
u = A.Uat978()
t = time.time()
for seed in range(100, 100 + n2):
    T2.test_fuzzed_phase_streams(u, seed)
print("978: %d fuzz seeds identical to the oracle (%.0f s), extra look-ups %d" % (n2, time.time() - t, u.timing()["extra_lookups"]), flush=True)
