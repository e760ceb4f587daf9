"""First-contact probe on the GPU box: which load order of the HIP runtime works, basic scan sanity."""
import subprocess
import sys

SNIPPETS = {
    "torch_first": "import torch; print('torch', torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0));\n",
    "lib_first": "",
}
BODY = r"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import libadsb_amd as A
from libadsb_amd import synth
import helpers as H
s = A.Scanner()
iq, inj = synth.fill_range(0, 2)
t = time.time(); got = s.scan(iq, A.REF_BUFFER_BYTES); dt = time.time() - t
want = H.expected_records(iq, A.REF_BUFFER_BYTES)
print('records', len(got), 'expected', len(want), 'equal', got.shape == want.shape and bool(np.array_equal(got, want)), 'first scan s', round(dt, 3))
if not (got.shape == want.shape and np.array_equal(got, want)):
    n = min(len(got), len(want))
    bad = [i for i in range(n) if got[i] != want[i]][:5]
    for i in bad: print(' diff', i, got[i], want[i])
    gk = set((int(r['offset']), int(r['flags']) & 1) for r in got[got['buffer'] == 0]); wk = set((int(r['offset']), int(r['flags']) & 1) for r in want[want['buffer'] == 0])
    print(' only gpu', sorted(gk - wk)[:10], 'only oracle', sorted(wk - gk)[:10])
m = s.magnitude(iq[:4096]); from oracle import oracle_py as O
print('magnitude equal', bool(np.array_equal(m, O.magnitude(iq[:4096]))))
"""
for name, pre in SNIPPETS.items():
    try:
        r = subprocess.run([sys.executable, "-c", pre + BODY], capture_output=True, text=True, timeout=300)
        print("==", name, "rc", r.returncode)
        print(r.stdout[-3000:])
        print(r.stderr[-3000:])
    except subprocess.TimeoutExpired:
        print("==", name, "TIMEOUT")
        break
