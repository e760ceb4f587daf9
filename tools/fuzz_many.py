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

# generator-driven: random generator settings, random call sizes
import numpy as np  # noqa: E402
from libadsb_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
import helpers as H  # noqa: E402

n3 = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rng = np.random.default_rng(7)
BB = A.REF_BUFFER_BYTES
t = time.time()
for k in range(n3):
    cfg = synth.default_cfg(noise_amp=int(rng.integers(0, 50)), mean_spacing=int(rng.choice([0, 250, 600, 2000, 20000])),
                            amp_lo=int(rng.integers(5, 100)), amp_hi=int(rng.integers(100, 129)), pct_df17=int(rng.integers(0, 60)),
                            pct_df11=int(rng.integers(0, 40)), pct_bitflip=int(rng.integers(0, 100)), pct_halfsample=int(rng.integers(0, 100)))
    first = int(rng.integers(0, 10**6))
    iq, _ = synth.fill_range(first, 3, cfg=cfg)
    try:
        H.assert_records_equal(sc.scan(iq, BB), H.expected_records(iq, BB))
    except AssertionError:
        print("FAILED at k=%d first_buffer=%d cfg=%s" % (k, first, {f[0]: getattr(cfg, f[0]) for f in cfg._fields_}), flush=True)
        raise
    h = A.Handler1090()
    fr, ac = h.handle_data(iq, BB)
    ofr, oac = H.oracle_run(iq, BB)
    H.assert_streams_equal(fr, ac, ofr, oac)
    h.close()
print("1090: %d random generator settings identical to the oracle (%.0f s)" % (n3, time.time() - t), flush=True)
t = time.time()
for k in range(n3):
    cfg = synth.default_cfg978(noise_amp=int(rng.integers(0, 30)), amp_lo=int(rng.integers(3, 40)), amp_hi=int(rng.integers(40, 120)),
                               mean_gap_bits=int(rng.choice([0, 30, 300, 3000, 50000])), pct_uplink=int(rng.integers(0, 100)),
                               pct_long=int(rng.integers(0, 100)), pct_corrupt=int(rng.integers(0, 100)), max_bad_bytes=int(rng.integers(1, 16)))
    iq = synth.fill978(int(rng.integers(0, 10**6)), 3 * BB, cfg)
    full = bool(rng.integers(0, 2))
    uu, oo = A.Uat978(carry_full=full), O.Oracle978(carry_full=full)
    pos = 0
    while pos < iq.size:
        n = int(rng.choice([BB, BB, 2 * BB, 65536, 30000, 100002]))
        part = iq[pos:pos + n]
        assert uu.handle_data(part) == oo.handle_data(part), (k, pos)
        assert uu.stream_state() == oo.stream_state()
        pos += n
    uu.close()
print("978: %d random generator settings / call sizes identical to the oracle (%.0f s)" % (n3, time.time() - t), flush=True)

# the 2.4 MS/s mode against its specification, random generator settings
sc24 = A.Scanner(mode=A.MODE_2400)
t = time.time()
for k in range(n3):
    cfg = synth.default_cfg(noise_amp=int(rng.integers(0, 40)), mean_spacing=int(rng.choice([0, 250, 600, 2000, 20000])),
                            amp_lo=int(rng.integers(5, 100)), amp_hi=int(rng.integers(100, 129)), pct_df17=int(rng.integers(0, 60)),
                            pct_df11=int(rng.integers(0, 40)), pct_bitflip=int(rng.integers(0, 100)))
    iq, _ = synth.fill_range(int(rng.integers(0, 10**6)), 3, cfg=cfg, rate_x10=24)
    got = sc24.scan(iq, BB)
    want = O.expected_records2400(iq, BB, dtype=A.RECORD_DTYPE)
    assert len(got) == len(want) and got.tobytes() == want.tobytes(), ("mode 2400", k)
    if k % 20 == 19:
        print("  2400: %d settings so far" % (k + 1), flush=True)
print("2400: %d random generator settings identical to the specification (%.0f s)" % (n3, time.time() - t), flush=True)
