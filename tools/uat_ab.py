"""A/B of library builds on the UAT 978 workload: one bench.py process per build (ADSB_AMD_LIB), the same synthetic GiB.
    python tools/uat_ab.py ab_ship/a.so ab_ship/b.so ...
"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sys.argv[1:]:
    env = dict(os.environ, ADSB_AMD_LIB=os.path.join(root, lib))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "uat978", "--steps", os.environ.get("AB_STEPS", "20"), "--warmup", "3", "--cpu-buffers", "0"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
    d = json.loads(out.strip().splitlines()[-1])
    print("%-22s step %.4f ms  serial %.4f  scan %.4f  demod %.4f  host %s" % (lib, d["ms_per_step"], d["ms_per_step_serial"], d["roofline"]["kernel_ms"],
                                                                            d["demod_kernel_ms"], {k: round(v, 3) for k, v in d["host_wall_ms_last_step"].items()}), flush=True)
