"""A/B of library builds on the recorded-file replay (adsb_amd_handler_replay_file): python tools/replay_ab.py a.so b.so ...
One process per library (ADSB_AMD_LIB), the same synthetic GiB written to /dev/shm once, a new handler per run as a caller would have."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = "/dev/shm/adsb_amd_replay_ab.test.dat"


def child():
    sys.path.insert(0, ROOT)
    import libadsb_amd as A
    ts = []
    n = None
    for _ in range(int(os.environ.get("REPLAY_RUNS", "7"))):
        h = A.Handler1090()
        t = time.perf_counter()
        n, _, _ = h.replay_file(PATH, collect=False)
        ts.append(time.perf_counter() - t)
        h.close()
    ts.sort()
    print(json.dumps({"lib": os.path.basename(A.LIB_PATH), "accepted": int(n), "best_ms": round(ts[0] * 1e3, 2), "median_ms": round(ts[len(ts) // 2] * 1e3, 2),
                      "all_ms": [round(t * 1e3, 1) for t in ts]}))


def main():
    sys.path.insert(0, ROOT)
    from libadsb_amd import synth
    if not os.path.exists(PATH):
        iq, _ = synth.fill_range(0, 4096)
        iq.tofile(PATH)
    try:
        for rnd in range(2):
            for lib in sys.argv[1:]:
                env = dict(os.environ, ADSB_AMD_LIB=os.path.abspath(lib), REPLAY_CHILD="1")
                out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
                line = [l for l in out.stdout.splitlines() if l.startswith("{")]
                print(line[-1] if line else ("FAILED " + lib + ": " + out.stderr[-800:]), flush=True)
    finally:
        os.unlink(PATH)


if __name__ == "__main__":
    child() if os.environ.get("REPLAY_CHILD") else main()
