"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

import libadsb_amd as A
from libadsb_amd import synth
from oracle import oracle_py as O

import helpers as H
from libadsb_amd.shard import shard_range

pytestmark = pytest.mark.gpu
BB = A.REF_BUFFER_BYTES


@pytest.fixture(scope="module")
def scanner(native_libs):
    s = A.Scanner()
    yield s
    s.close()


def test_magnitude_all_iq_pairs(scanner):
    # every (I, Q) byte pair once: the whole domain of the reference's magnitude LUT (ADSB1090.cpp:131-142, 165-173)
    i, q = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    iq = np.stack([i.ravel(), q.ravel()], axis=1).ravel()
    assert np.array_equal(scanner.magnitude(iq), O.magnitude(iq))


@pytest.mark.parametrize("over", [
    dict(),
    dict(noise_amp=20),
    dict(mean_spacing=300, noise_amp=8),
    dict(amp_lo=100, amp_hi=128, noise_amp=1, pct_halfsample=50),
    dict(pct_bitflip=100, pct_df17=50, pct_df11=50),
    dict(mean_spacing=0, noise_amp=40),
    dict(noise_amp=0, mean_spacing=600, amp_lo=3, amp_hi=12, pct_halfsample=40),   # frames around the energy gate (mean |lo-hi| 1275)
    dict(noise_amp=1, mean_spacing=400, amp_lo=5, amp_hi=9, pct_bitflip=30),
])
def test_records_match_oracle(scanner, over):
    cfg = synth.default_cfg(**over)
    iq, _ = synth.fill_range(3, 6, cfg=cfg)
    got = scanner.scan(iq, BB)
    want = H.expected_records(iq, BB)
    H.assert_records_equal(got, want)
    if over.get("mean_spacing", 1) != 0:
        assert len(got) > 100


def test_whole_input_as_one_buffer(scanner):
    # the reference's TestEmbedded feeds a whole file through one HandleData call (tests/test_1090.cpp:66-67)
    iq, _ = synth.fill_range(40, 5)
    H.assert_records_equal(scanner.scan(iq, 0), H.expected_records(iq, 0))


@pytest.mark.parametrize("nbytes", [0, 2, 478, 480, 482, 496, 8190, 8192 + 480, 8192 + 482, 16384 + 494, 3 * 8192 + 2 * 250, 262144 - 2, 262144 + 6])
def test_ragged_single_buffer(scanner, nbytes):
    base, _ = synth.fill_range(7, 2, cfg=synth.default_cfg(mean_spacing=400))
    iq = base[:nbytes]
    H.assert_records_equal(scanner.scan(iq, 0), H.expected_records(iq, 0))


def test_trailing_partial_buffer_is_ignored(scanner):
    iq, _ = synth.fill_range(11, 3)
    H.assert_records_equal(scanner.scan(iq[:2 * BB + 1000], BB), H.expected_records(iq[:2 * BB], BB))


def test_frame_at_offset_zero_and_at_buffer_end(scanner):
    # j == 0 skips the m[-1] read and the phase retry (ADSB1090.cpp:820); a frame starting inside the last 240
    # samples of a buffer is invisible (loop bound :772)
    cfg = synth.default_cfg(mean_spacing=0)
    iq, _ = synth.fill_range(0, 1, cfg=cfg)
    src, man = synth.fill(5, cfg=synth.default_cfg(noise_amp=0, mean_spacing=3000), manifest=True)
    f = man[0]
    seg = src[2 * int(f["start"]): 2 * (int(f["start"]) + 250)].copy()
    n = BB // 2
    for at in (0, 1, n - 241, n - 240, n - 239, 4095, 4096, 4097, 8191):
        buf = iq.copy()
        k = min(250, n - at)
        buf[2 * at: 2 * (at + k)] = seg[:2 * k]
        H.assert_records_equal(scanner.scan(buf, BB), H.expected_records(buf, BB))


def test_uniform_random_bytes_overflow_path(scanner):
    # adversarial input: full-scale noise produces dense candidates and exercises the per-chunk region growth
    rng = np.random.default_rng(1090)
    iq = rng.integers(0, 256, size=BB, dtype=np.uint8)
    H.assert_records_equal(scanner.scan(iq, BB), H.expected_records(iq, BB))


def test_constant_inputs(scanner):
    for v in (0, 127, 128, 255):
        iq = np.full(BB, v, dtype=np.uint8)
        assert len(scanner.scan(iq, BB)) == 0


def test_handler_callback_stream_matches_oracle(native_libs):
    # HandleData semantics end to end: one call per 262144-byte buffer, state carried across calls
    iq, _ = synth.fill_range(0, 12)
    h = A.Handler1090()
    o = O.Oracle1090()
    for b in range(12):
        chunk = iq[b * BB:(b + 1) * BB]
        fr, ac = h.handle_data(chunk)
        ofr, oac = o.handle_data(chunk)
        H.assert_streams_equal(fr, ac, ofr, oac)
        assert H.callback_text(ac) == H.callback_text(oac)
    h.close()


def test_handler_multi_buffer_call_equals_per_buffer_calls(native_libs):
    iq, _ = synth.fill_range(100, 10)
    h = A.Handler1090()
    fr, ac = h.handle_data(iq, BB)
    ofr, oac = H.oracle_run(iq, BB)
    H.assert_streams_equal(fr, ac, ofr, oac)
    h.close()


def test_large_shard_invariance_and_determinism(scanner):
    # size-independent properties at a size the oracle is not run on: scanning 256 buffers at once equals the
    # concatenation of two half scans (buffers are independent), and a repeat is bit-identical
    nbuf = 256
    iq, injected = synth.fill_range(1000, nbuf)
    full = scanner.scan(iq, BB)
    again = scanner.scan(iq, BB)
    assert np.array_equal(full, again)
    a = scanner.scan(iq[:nbuf // 2 * BB], BB)
    b = scanner.scan(iq[nbuf // 2 * BB:], BB)
    b["buffer"] += nbuf // 2
    H.assert_records_equal(full, np.concatenate([a, b]))
    key = full["buffer"].astype(np.uint64) * (1 << 33) + full["offset"].astype(np.uint64) * 2 + (full["flags"] & 1)
    assert np.all(np.diff(key.astype(np.int64)) > 0), "records must be strictly sorted by (buffer, offset, pass)"
    stateless = full[(full["flags"] & A.F_NEEDS_ICAO) == 0]
    assert 0.6 * injected < len(stateless) < injected
    # every unconditional record carries a message whose parity is consistent (checksum of checksums)
    for r in stateless[:: max(1, len(stateless) // 500)]:
        msg = bytes(r["msg"])
        nb = int(r["nbits"])
        stored = int.from_bytes(msg[nb // 8 - 3: nb // 8], "big")
        assert O.lib().oracle1090_checksum(msg, nb) == stored


def test_async_two_slot_pipeline(scanner):
    import torch
    iq0, _ = synth.fill_range(0, 8)
    iq1, _ = synth.fill_range(8, 8)
    d0 = torch.from_numpy(iq0).cuda()
    d1 = torch.from_numpy(iq1).cuda()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    scanner.submit(d0.data_ptr(), d0.numel(), BB, st, 0)
    scanner.submit(d1.data_ptr(), d1.numel(), BB, st, 1)
    r0 = scanner.fetch(0)
    r1 = scanner.fetch(1)
    H.assert_records_equal(r0, scanner.scan(iq0, BB))
    H.assert_records_equal(r1, scanner.scan(iq1, BB))
    k_ms, t_ms = scanner.timing(1)
    assert 0 < k_ms <= t_ms


def test_timing_events_on_every_nth_scan(native_libs):
    """adsb_amd_set_timing: the two events that time the demodulation kernel ride on every n-th submit; a scan without them reports no time."""
    sc = A.Scanner()
    iq, _ = synth.fill_range(3, 4)
    want = sc.scan(iq, BB)
    assert sc.timing_if_timed(0) is not None  # the default: every scan
    sc.set_timing(2)
    seen = []
    for _ in range(4):
        H.assert_records_equal(sc.scan(iq, BB), want)
        seen.append(sc.timing_if_timed(0))
    assert [t is not None for t in seen] == [True, False, True, False] and all(0 < t[0] <= t[1] for t in seen if t)
    with pytest.raises(A.AdsbAmdError, match="timed"):
        sc.timing(0)
    sc.set_timing(0)
    sc.scan(iq, BB)
    assert sc.timing_if_timed(0) is None
    sc.close()


def test_cxx_drop_in_handler_matches_reference_test_flow(native_libs, tmp_path):
    # the reference's own test shape (tests/test_1090.cpp): C++ listener, ADSB::test::TryCreateADSB1090Handler, HandleData,
    # one formatted line per OnChanged; here against the oracle's callback text instead of the (input-less) goldens
    import subprocess
    from libadsb_amd import build
    exe = build.build_cxx_test()
    iq, _ = synth.fill_range(200, 6)
    path = tmp_path / "iq1090.bin"
    iq.tofile(path)
    for bb in (BB, 0):
        out = subprocess.run([exe, str(path), str(bb)], capture_output=True, timeout=120)
        assert out.returncode == 0, out.stderr.decode()
        got = [l for l in out.stdout.split(b"\n") if l and not l.startswith(b"/opt/amdgpu")]
        o = O.Oracle1090(sample_clock_hz=0)  # wall clock, like the C++ handler and the reference
        _, oac = H.oracle_run(iq, bb, oracle=o)
        want = [t.encode("latin-1") for t in H.callback_text(oac)]
        assert got == want
        assert len(got) > 100


def test_replay_file_matches_oracle_and_drops_the_partial_buffer(native_libs, tmp_path):
    """Recorded-file replay (RTLSDR.hpp:419-442): whole 262144-B buffers in order, the trailing partial read is never
    delivered; a rank's sub-range equals the same buffers handled alone."""
    iq, _ = synth.fill_range(40, 6)
    path = tmp_path / "1090000000.test.dat"
    with open(path, "wb") as f:
        f.write(iq.tobytes())
        f.write(iq[:100001].tobytes())  # partial seventh buffer
    h = A.Handler1090()
    n, fr, ac = h.replay_file(str(path))
    ofr, oac = H.oracle_run(iq, BB)
    assert n == len(fr) == len(ofr) > 100
    H.assert_streams_equal(fr, ac, ofr, oac)
    h.close()
    h = A.Handler1090()
    n2, fr2, ac2 = h.replay_file(str(path), first_buffer=2, max_buffers=3)
    ofr2, oac2 = H.oracle_run(iq[2 * BB:5 * BB], BB)
    assert n2 == len(ofr2)
    H.assert_streams_equal(fr2, ac2, ofr2, oac2)
    assert h.replay_file(str(path), first_buffer=6)[0] == 0
    with pytest.raises(A.AdsbAmdError, match="cannot open"):
        h.replay_file(str(tmp_path / "missing.dat"))
    h.close()


def test_replay_file_over_many_batches_and_twice_on_one_handler(native_libs, tmp_path):
    """The replay's stages (readers -> uploader -> scan -> resolver, capi.cpp) keep four page-locked and three device staging buffers of 64
    reference buffers each: a 330-buffer file (six batches, the last one partial) goes round every ring, and the callback stream must still be
    the oracle's for the whole file in file order.  The staging is the handler's: a second pass on the same handler re-uses it, and -- like the
    reference, whose handler keeps its aircraft and ICAO state while RTLSDR re-opens the file -- continues the same stream."""
    nbuf = 330
    iq, _ = synth.fill_range(7, nbuf)
    path = tmp_path / "1090000000.test.dat"
    iq.tofile(path)
    h = A.Handler1090()
    n, fr, ac = h.replay_file(str(path))
    ofr, oac = H.oracle_run(iq, BB)
    assert n == len(fr) == len(ofr) > 5000
    H.assert_streams_equal(fr, ac, ofr, oac)
    n2, fr2, ac2 = h.replay_file(str(path), first_buffer=100, max_buffers=200)  # four batches (the last partial) through the staging as it stands
    h.close()
    # one stream: the whole file, then buffers 100..299 again, through one handler state
    both = np.concatenate([iq, iq[100 * BB:300 * BB]])
    ofr_b, oac_b = H.oracle_run(both, BB)
    assert n + n2 == len(ofr_b)
    fr2 = fr2.copy()
    fr2["offset"] += nbuf * (BB // 2)  # a frame's offset counts from the first buffer of ITS pass (whatever the batches are): here 100 -> 0
    H.assert_streams_equal(np.concatenate([fr, fr2]), np.concatenate([ac, ac2]), ofr_b, oac_b)


def test_full_size_1gib_equals_oracle_on_every_buffer(scanner):
    """BASELINE configs[1]+[2] at full size: 4096 reference buffers (1 GiB) in one scan.  Size-independent properties
    (shard invariance, strict order, parity consistency) plus exact equality with the oracle on all 4096 buffers."""
    nbuf = 4096
    iq, injected = synth.fill_range(0, nbuf, nthreads=16)
    full = scanner.scan(iq, BB)
    parts = []
    for k in range(8):  # the 8-GPU partition of SURVEY.md 8(e), on one GPU
        lo, cnt = shard_range(nbuf, k, 8)
        part = scanner.scan(iq[lo * BB:(lo + cnt) * BB], BB)
        part["buffer"] += lo
        parts.append(part)
    H.assert_records_equal(full, np.concatenate(parts))
    key = full["buffer"].astype(np.uint64) * (1 << 33) + full["offset"].astype(np.uint64) * 2 + (full["flags"] & 1)
    assert np.all(np.diff(key.astype(np.int64)) > 0)
    stateless = full[(full["flags"] & A.F_NEEDS_ICAO) == 0]
    assert 0.6 * injected < len(stateless) < injected
    for r in stateless[:: max(1, len(stateless) // 2000)]:
        msg, nb = bytes(r["msg"]), int(r["nbits"])
        assert O.lib().oracle1090_checksum(msg, nb) == int.from_bytes(msg[nb // 8 - 3: nb // 8], "big")
    # every one of the 4096 buffers against the oracle: the expected record array is built in C (oracle/expected1090.c,
    # the same rules as helpers.records_from_probe, buffers spread over threads)
    H.assert_records_equal(full, O.expected_records(iq, BB, dtype=A.RECORD_DTYPE))


@pytest.mark.parametrize("depth,nstreams", [(2, 1), (3, 1), (3, 2)])
def test_pipelined_loop_with_a_resubmit_beside_every_copy_delivers_each_steps_own_records(native_libs, depth, nstreams):
    """The loop bench.py runs, on inputs that differ from step to step: two scans on the stream at any time, a step's ordering pass in front of
    the scan kernel after it (gather1090.hip.h), its count stored by the pass's last finisher, its packed records copied while the slot's NEXT scan
    is already submitted (adsb_amd_scan_1090_fetch_packed_begin / _end).  Every step's records must be that step's own -- a copy that started before
    the pass had finished, or a pass that wrote into an array still being copied, shows as records of another input -- and equal to what the same
    scanner delivers for the same input on its own (serial submit / fetch: the stand-alone pass), which the other tests hold against the oracle.
    Sizes: 1 024 buffers (256 MiB, 128 blocks of the ordering pass) and 40 buffers (a pass of 5 blocks inside a kernel of 1 280 waves).
    With two scans on the stream (bench.py's loop), with three (the context has three result slots), and with three alternating over TWO streams
    (bench.py --streams 2 --depth 3: a scan kernel starts while the one before it drains, a step's ordering pass rides in the kernel two steps on)."""
    import torch
    side = [torch.cuda.Stream() for _ in range(nstreams - 1)]
    for nbuf, steps in ((1024, 9), (40, 7)):
        sc = A.Scanner()
        sc.set_outputs(A.OUT_PACKED)
        st = torch.cuda.current_stream().cuda_stream
        sts = [st] + [t.cuda_stream for t in side]
        inputs, want = [], []
        for k in range(3):
            iq, _ = synth.fill_range(7000 + k * nbuf, nbuf, nthreads=16)
            d = torch.from_numpy(iq).cuda()
            inputs.append(d)
            sc.submit(d.data_ptr(), d.numel(), BB, st, 0)
            want.append(sc.fetch_packed(0, copy=True))
        assert len({len(w) for w in want}) == 3  # (the three inputs hold different numbers of records: a mix-up cannot pass by count)
        got = []
        torch.cuda.synchronize()
        for k in range(depth):
            sc.submit(inputs[k % 3].data_ptr(), inputs[k % 3].numel(), BB, sts[k % nstreams], k)
        for i in range(steps):
            n = sc.fetch_packed_begin(i % depth)
            if i + depth < steps:
                d = inputs[(i + depth) % 3]
                sc.submit(d.data_ptr(), d.numel(), BB, sts[(i + depth) % nstreams], i % depth)
            rec = sc.fetch_packed_end(i % depth, copy=True)
            assert len(rec) == n
            got.append(rec)
        for i, rec in enumerate(got):
            w = want[i % 3]
            assert len(rec) == len(w) and rec.tobytes() == w.tobytes(), "step %d of the pipelined loop (%d buffers) differs from the same input scanned on its own" % (i, nbuf)
        sc.close()


def test_split_fetch_refuses_calls_out_of_order(native_libs):
    """adsb_amd_scan_1090_fetch_packed_begin / _end: a begin without a submit, a second begin before the end, an end without a begin, a slot that
    does not exist and a scan that did not produce the packed form are refused with a state / argument error and leave the context usable."""
    import torch
    iq, _ = synth.fill_range(90, 8)
    d = torch.from_numpy(iq).cuda()
    st = torch.cuda.current_stream().cuda_stream
    sc = A.Scanner()
    sc.set_outputs(A.OUT_PACKED)
    with pytest.raises(A.AdsbAmdError, match="fetch without submit"):
        sc.fetch_packed_begin(0)
    with pytest.raises(A.AdsbAmdError, match="no fetch begun"):
        sc.fetch_packed_end(0)
    with pytest.raises(A.AdsbAmdError, match="slot must be"):
        sc.fetch_packed_begin(A.SLOTS if hasattr(A, "SLOTS") else 3)
    sc.submit(d.data_ptr(), d.numel(), BB, st, 1)
    n = sc.fetch_packed_begin(1)
    sc.submit(d.data_ptr(), d.numel(), BB, st, 1)  # the slot's next scan beside the copy: allowed, that is the point of the split
    with pytest.raises(A.AdsbAmdError, match="begun and not ended"):
        sc.fetch_packed_begin(1)
    first = sc.fetch_packed_end(1, copy=True)
    assert len(first) == n > 0
    with pytest.raises(A.AdsbAmdError, match="no fetch begun"):
        sc.fetch_packed_end(1)
    again = sc.fetch_packed(1, copy=True)  # the scan submitted beside the copy (the refused begin has not consumed it)
    assert again.tobytes() == first.tobytes()
    sc.set_outputs(A.OUT_RECORDS)
    sc.submit(d.data_ptr(), d.numel(), BB, st, 0)
    with pytest.raises(A.AdsbAmdError, match="did not produce the packed form"):
        sc.fetch_packed_begin(0)
    sc.set_outputs(A.OUT_PACKED)
    sc.submit(d.data_ptr(), d.numel(), BB, st, 0)
    assert sc.fetch_packed(0, copy=True).tobytes() == first.tobytes()
    sc.close()


@pytest.mark.parametrize("seed", range(5))
def test_fuzzed_threshold_cases(scanner, seed):
    """Buffers built to sit on the slicer's thresholds: samples from a small alphabet (many exact ties), frames whose pulse
    and gap levels differ by about the 256-magnitude "decided" threshold, strong pulses whose 5/4 phase correction wraps
    16 bits, pulses with the preceding sample high (out-of-phase preambles), frames back to back and across chunk edges."""
    rng = np.random.default_rng(1090 + seed)
    nb, n = 3, BB // 2
    out = np.empty((nb, n, 2), dtype=np.uint8)
    alphabet = np.array([[127, 127], [128, 126], [129, 127], [127, 130], [131, 131], [140, 127], [127, 150], [160, 160], [200, 127], [255, 255],
                         [0, 0], [255, 127], [127, 0], [120, 134]], dtype=np.uint8)
    for b in range(nb):
        base = alphabet[rng.integers(0, 5 if seed % 2 else len(alphabet), n)]
        out[b] = base
        pos = int(rng.integers(0, 50))
        while pos < n - 260:
            hi_level = alphabet[rng.integers(5, len(alphabet))]
            lo_level = alphabet[rng.integers(0, 6)]
            if rng.random() < 0.2:  # around the energy gate: a mean |lo-hi| of 1275 lies between amplitudes 3 (1080) and 4 (1440)
                hi_level = np.array([127 + int(rng.integers(3, 6)), 127 + int(rng.integers(0, 2))], dtype=np.uint8)
                lo_level = np.array([127, 127 + int(rng.integers(0, 2))], dtype=np.uint8)
            elif rng.random() < 0.4:  # levels about 256 apart in magnitude: |I-127| differs by ~0.7
                hi_level = np.array([127 + int(rng.integers(20, 60)), 127], dtype=np.uint8)
                lo_level = np.array([hi_level[0] - int(rng.integers(0, 3)), 127 + int(rng.integers(0, 2))], dtype=np.uint8)
            pulses = [0, 2, 7, 9]
            frame = out[b, pos:pos + 240]
            frame[:16] = lo_level
            for k in pulses:
                frame[k] = hi_level
            nbits = 112 if rng.random() < 0.6 else 56
            bits = rng.integers(0, 2, 112)
            df = int(rng.choice([17, 11, 4, 20, 0, 5, 21, 16, 24, 19, 1]))
            bits[:5] = [(df >> (4 - k)) & 1 for k in range(5)]
            for k in range(nbits):
                first, second = (hi_level, lo_level) if bits[k] else (lo_level, hi_level)
                if rng.random() < 0.08:
                    second = first  # a tie inside the frame
                frame[16 + 2 * k], frame[17 + 2 * k] = first, second
            if rng.random() < 0.3 and pos > 0:
                out[b, pos - 1] = hi_level  # energy before the preamble: DetectOutOfPhase
            pos += int(rng.choice([240, 241, 130, 500, 4096 - 120, 2000]))
    iq = out.reshape(-1)
    got = scanner.scan(iq, BB)
    want = H.expected_records(iq, BB)
    H.assert_records_equal(got, want)
    # and through the handler: same accepted frames, aircraft snapshots and callback text as the oracle's sequential loop
    h = A.Handler1090()
    fr, ac = h.handle_data(iq, BB)
    ofr, oac = H.oracle_run(iq, BB)
    H.assert_streams_equal(fr, ac, ofr, oac)
    h.close()


def test_expected_records_in_c_equal_the_python_rules():
    # the two statements of the emission contract (tests/helpers.py and oracle/expected1090.c) agree, including on inputs
    # that are all ties / all noise
    for over in (dict(), dict(noise_amp=20), dict(pct_bitflip=100, pct_df17=50, pct_df11=50), dict(noise_amp=0, mean_spacing=600, amp_lo=3, amp_hi=12)):
        iq, _ = synth.fill_range(17, 3, cfg=synth.default_cfg(**over))
        H.assert_records_equal(O.expected_records(iq, BB, dtype=A.RECORD_DTYPE), H.expected_records(iq, BB))
        H.assert_records_equal(O.expected_records(iq[:BB + 9000], 0, dtype=A.RECORD_DTYPE), H.expected_records(iq[:BB + 9000], 0))


def test_handle_data_from_another_thread(native_libs):
    """libadsb calls HandleData from its transport's consumer thread (RTLSDR.hpp:470-473), not from the thread that built the
    handler; HIP's current device is per thread, so the entry point has to select the handler's device itself."""
    import threading
    iq, _ = synth.fill_range(60, 4)
    h = A.Handler1090(0)
    got, err = {}, []

    def consumer():
        try:
            for b in range(4):
                fr, ac = h.handle_data(iq[b * BB:(b + 1) * BB])
                got[b] = (fr, ac)
        except Exception as e:  # surfaced on the main thread
            err.append(e)
    t = threading.Thread(target=consumer)
    t.start()
    t.join()
    assert not err, err
    o = O.Oracle1090()
    for b in range(4):
        ofr, oac = o.handle_data(iq[b * BB:(b + 1) * BB])
        H.assert_streams_equal(got[b][0], got[b][1], ofr, oac)
    # and the replay entry point, which goes through the same staging path
    h.close()


def test_fetch_device_delivers_the_same_records(scanner):
    import torch
    iq, _ = synth.fill_range(300, 8)
    d = torch.from_numpy(iq).cuda()
    want = scanner.scan(iq, BB)
    dst = torch.zeros((len(want) + 100) * 32, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    scanner.submit(d.data_ptr(), d.numel(), BB, st, 1)
    n = scanner.fetch_device(1, dst.data_ptr(), len(want) + 100, st)
    torch.cuda.synchronize()
    assert n == len(want)
    H.assert_records_equal(dst[:n * 32].cpu().numpy().view(A.RECORD_DTYPE), want)
    scanner.submit(d.data_ptr(), d.numel(), BB, st, 1)
    with pytest.raises(A.AdsbAmdError, match="too small"):
        scanner.fetch_device(1, dst.data_ptr(), 10, st)
    H.assert_records_equal(scanner.scan(iq, BB), want)  # the slot is usable again after the failed delivery


def _run(cmd, timeout=900):
    import os
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_sharded_file_step_two_ranks_rehearsal(native_libs, tmp_path):
    """BASELINE configs[3] as bench.py runs it for --gpus N > 1, rehearsed with two ranks on this one GPU (gloo): started bare
    (bench.py launches its own ranks), rank r scans its half of the recording, the records are gathered on rank 0 and resolved
    there.  The gathered records and the callback stream must equal the oracle run over the whole recording."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = tmp_path / "stream.npz"
    out = _run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--mib", "4", "--steps", "3", "--warmup", "1",
                "--dump-stream", str(dump)])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["buffers_total"] == 32 and line["value"] > 0
    z = np.load(dump)
    whole, _ = synth.fill_range(0, 32)
    # what travels is the packed form (record head + GPU-decoded fields, no message bytes): it must be the oracle's records and the
    # host build's decoded fields, merged
    want = O.expected_records(whole, BB, dtype=A.RECORD_DTYPE)
    assert z["records"].dtype == A.PACKED_DTYPE and np.array_equal(z["records"], A.pack_records(want, A.decode_records_host(want)))
    ofr, oac = H.oracle_run(whole, BB)
    ofr = ofr.copy()
    ofr["msg"] = 0  # frames resolved from the packed form carry every field but the message bytes
    H.assert_streams_equal(z["frames"], z["aircraft"], ofr, oac)
    assert line["decoded_msgs_per_step"] == len(ofr) and line["ranks_seen"] == 2 and line["end_to_end_msamples_per_s"] > 0
    assert len(line["kernel_ms_by_rank"]["all"]) == 2
    # round 6: the rehearsal takes the production record path -- each of the two processes registers its own segments of the shared file and its
    # GPU context writes them, rank 0 reads both unregistered -- and the records gathered the other way (host side here) are the same bytes
    assert line["record_transport_by_rank"] == ["NodeGather", "NodeGather"] and line["record_transports_agree"] is True


def _gpu_count():
    import torch
    return torch.cuda.device_count()  # (does not initialise the GPU in this process)


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: the N > 1 step over RCCL, one process per GPU")
def test_bench_sharded_file_step_two_gpus_over_rccl(native_libs, tmp_path):
    """BASELINE configs[3] for real whenever the suite lands on a box with more than one GPU: `bench.py --gpus 2` started bare (it
    launches its own two ranks, backend nccl = RCCL), rank r on GPU r scans its half of a 128 MiB recording, every GPU writes its packed
    records into its own page-locked segment of node-shared memory (a PEER process's GPU writing memory rank 0's CPU reads: what a
    world of one cannot show), the headers travel through the control page.  Delivered records and the resolved callback stream must
    equal the oracle run over the whole recording, both record transports must agree."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = tmp_path / "stream2.npz"
    out = _run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mib", "64", "--steps", "4", "--warmup", "1", "--dump-stream", str(dump)])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    nbuf = 2 * (64 << 20) // BB
    assert line["n_gpus"] == 2 and line["config"]["buffers_total"] == nbuf and line["value"] > 0
    assert "RCCL" in line["config"]["workload"] and line["config"]["record_transport"] == "node-shared page-locked host segments"
    z = np.load(dump)
    whole, _ = synth.fill_range(0, nbuf)
    want = O.expected_records(whole, BB, dtype=A.RECORD_DTYPE)
    assert z["records"].dtype == A.PACKED_DTYPE and np.array_equal(z["records"], A.pack_records(want, A.decode_records_host(want)))
    ofr, oac = H.oracle_run(whole, BB)
    ofr = ofr.copy()
    ofr["msg"] = 0  # frames resolved from the packed form carry every field but the message bytes
    H.assert_streams_equal(z["frames"], z["aircraft"], ofr, oac)
    assert line["decoded_msgs_per_step"] == len(ofr) and line["ranks_seen"] == 2 and line["record_transports_agree"] is True
    assert len(line["kernel_ms_by_rank"]["all"]) == 2


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: one UAT stream cut over two ranks, the loop's position over RCCL send / recv")
def test_bench_uat_one_stream_cut_over_two_gpus(native_libs):
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = _run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "uat978", "--mib", "64", "--steps", "3", "--warmup", "1"])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    one = line["one_stream_cut_over_ranks"]
    assert line["n_gpus"] == 2 and one["frames"] > 0 and one["equals_one_call"] is True, one


NCCL_WORLD1 = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import libadsb_amd as A
from libadsb_amd import synth
from libadsb_amd.shard import RootGather
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
BB = A.REF_BUFFER_BYTES
iq, _ = synth.fill_range(5, 8)
d = torch.from_numpy(iq).cuda()
sc = A.Scanner(0)
want = sc.scan(iq, BB)
rg = RootGather(len(want) + 64)
st = torch.cuda.current_stream()
for first in (0, 40):
    sc.submit(d.data_ptr(), d.numel(), BB, st.cuda_stream, 0)
    n = sc.fetch_device(0, rg.records_ptr(), rg.cap, st.cuda_stream)
    rec = rg.gather(n, first)
    w = want.copy(); w["buffer"] += first
    assert rec.dtype == w.dtype and np.array_equal(rec, w), "gathered records differ"
print("NCCL-OK", len(want))
# the node-local hand-over: the GPU writes the records into a registered /dev/shm mapping, only the header is gathered
from libadsb_amd.shard import NodeGather
ng = NodeGather(len(want) + 64)
comm = torch.cuda.Stream()
for step, first in enumerate((0, 40, 7, 0, 3)):
    sc.submit(d.data_ptr(), d.numel(), BB, st.cuda_stream, step & 1)
    with torch.cuda.stream(comm):
        n = sc.fetch_device(step & 1, ng.records_ptr(step), ng.cap, comm.cuda_stream)
        parts = ng.gather(step, n, first)
    assert len(parts) == 1 and parts[0][1] == first
    rec = NodeGather.concatenate(parts)
    w = want.copy(); w["buffer"] += first
    assert rec.dtype == w.dtype and np.array_equal(rec, w), "records handed over through node-shared memory differ"
ng.close()
# the packed form through the same hand-over (what bench.py ships)
sc.set_outputs(A.OUT_PACKED)
sc.submit(d.data_ptr(), d.numel(), BB, st.cuda_stream, 0)
pk_want = sc.fetch_packed(0)
assert np.array_equal(pk_want, A.pack_records(want, A.decode_records_host(want)))
ng = NodeGather(len(want) + 64, dtype=A.PACKED_DTYPE, tag="packed")
for step in range(3):
    ng.acquire(step)
    sc.submit(d.data_ptr(), d.numel(), BB, st.cuda_stream, step & 1)
    with torch.cuda.stream(comm):
        n = sc.fetch_device(step & 1, ng.records_ptr(step), ng.cap, comm.cuda_stream, packed=True)
        parts = ng.gather(step, n, 0)
    assert parts[0][0].dtype == A.PACKED_DTYPE and np.array_equal(parts[0][0], pk_want)
    ng.release(step)
ng.close()
print("NODE-OK", len(want))
dist.destroy_process_group()
"""


def test_root_gather_over_rccl_world_of_one(native_libs, tmp_path):
    # the device path of the record gather (scanner -> send buffer -> dist.gather on "nccl" = RCCL -> page-locked host) with the
    # one rank this box has; more ranks on one GPU are refused by RCCL, the N = 2 logic is covered by the gloo rehearsal above
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "nccl1.py"
    script.write_text(NCCL_WORLD1 % {"root": root})
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29533",
                str(script)])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert "NCCL-OK" in out.stdout and "NODE-OK" in out.stdout


def test_sharded_step_with_a_world_of_one_over_rccl(native_libs):
    """bench.py's N > 1 step (scan, hand-over of the packed records through node-shared page-locked memory with its credits, the step's
    header through the control page) run once with one rank over RCCL: the rank is seen through the control page and by the collective
    backend, and both record transports deliver the same records.  What the sharded step costs beside the plain one
    (`sharded_over_plain`, 0.95 in round 6) is a figure of the bench line; the suite only holds it above 0.85."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port",
                "29541", os.path.join(root, "bench.py"), "--gpus", "1", "--sharded-step-on-one-rank", "--steps", "40", "--warmup", "5"])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["ranks_seen"] == 1 and line["rccl_ranks_seen"] == 1 and line["record_transports_agree"] in (True, None)
    # a loose gate (round 6 measures 0.947 - 0.956 on one box, three runs -- the plain loop it is held against got faster: it now resubmits a slot beside
    # its copy; round 4: 0.989, profiles/r04_sharded_world_of_one.txt; boxes and contexts differ by a few per cent): a collective or a wait put back into
    # the per-step path costs tens of per cent and fails here
    assert line["sharded_over_plain"] is not None and line["sharded_over_plain"] >= 0.85, line["sharded_over_plain"]


def test_device_field_decoder_equals_the_host_build(scanner):
    """The ordering pass decodes every record's fields on the GPU (decode1090.h).  Device == host build on: the records of a real
    scan (as delivered beside the records), every (east-west, north-south) velocity pair -- 4.2 million headings, the one place
    where two math libraries meet --, and records with random bytes."""
    import torch
    iq, _ = synth.fill_range(500, 16)
    d = torch.from_numpy(iq).cuda()
    st = torch.cuda.current_stream().cuda_stream
    scanner.submit(d.data_ptr(), d.numel(), BB, st, 0)
    rec, dec = scanner.fetch_decoded(0)
    H.assert_records_equal(rec, scanner.scan(iq, BB))
    assert len(rec) > 1000 and np.array_equal(dec, A.decode_records_host(rec))
    assert set(np.unique(dec["kind"])) == {0, 1, 2, 3, 4}
    # every velocity pair, checked against the reference's expression evaluated with numpy's (= the host's libm) atan2
    e, n = np.meshgrid(np.arange(-1023, 1024), np.arange(-1023, 1024), indexing="ij")
    e, n = e.ravel(), n.ravel()
    vel = np.zeros(e.size, dtype=A.RECORD_DTYPE)
    vel["df"], vel["nbits"] = 17, 112
    m = vel["msg"]
    m[:, 0], m[:, 4] = 17 << 3, (19 << 3) | 1
    m[:, 5] = np.where(e < 0, 4, 0) | ((np.abs(e) >> 8) & 3)
    m[:, 6] = np.abs(e) & 0xFF
    m[:, 7] = np.where(n < 0, 0x80, 0) | ((np.abs(n) >> 3) & 0x7F)
    m[:, 8] = (np.abs(n) & 7) << 5
    got = scanner.decode(vel)
    speed = np.floor(np.sqrt((n * n + e * e).astype(np.float64))).astype(np.int64)
    hd = np.trunc(np.arctan2(e.astype(np.float64), n.astype(np.float64)) * 360 / (np.pi * 2)).astype(np.int64)
    hd = np.where(hd < 0, hd + 360, hd)
    hd = np.where(speed == 0, 0, hd)
    assert np.all(got["kind"] == A.K_VELOCITY) and np.array_equal(got["a"], speed) and np.array_equal(got["b"].astype(np.int64), hd)
    # a sample of those through the host build as well (it is a Python loop)
    pick = np.random.default_rng(5).choice(e.size, 3000, replace=False)
    assert np.array_equal(got[pick], A.decode_records_host(vel[pick]))
    rnd = np.zeros(20000, dtype=A.RECORD_DTYPE)
    rng = np.random.default_rng(9)
    rnd["msg"] = rng.integers(0, 256, size=(20000, 14), dtype=np.uint8)
    rnd["df"] = rng.choice([0, 4, 5, 11, 16, 17, 20, 21, 24, 17, 17], 20000)
    assert np.array_equal(scanner.decode(rnd), A.decode_records_host(rnd))


def test_resolver_on_gpu_decoded_fields_equals_oracle(scanner):
    # records + GPU-decoded fields through the host resolver == the oracle's sequential loop (frames, aircraft, callback text)
    import torch
    iq, _ = synth.fill_range(77, 24)
    d = torch.from_numpy(iq).cuda()
    scanner.submit(d.data_ptr(), d.numel(), BB, torch.cuda.current_stream().cuda_stream, 0)
    rec, dec = scanner.fetch_decoded(0)
    n, fr, ac = A.Resolver().feed(rec, BB // 2, 24, decoded=dec)
    ofr, oac = H.oracle_run(iq, BB)
    H.assert_streams_equal(fr, ac, ofr, oac)
    assert H.callback_text(ac) == H.callback_text(oac) and n == len(ofr) > 1000


def test_replay_through_the_ring_and_the_production_factory(native_libs, tmp_path):
    """libadsb's replay mode (RTLSDR.hpp:396-442): (1) adsb_amd_handler_run_replay -- the recording through the 16-slot ring, one
    HandleData per slot on the consumer thread -- gives the oracle's stream; (2) the PRODUCTION factory ADSB::TryCreateADSB1090Handler
    (ADSB.h:13-15), started the way DataProviderImpl starts it, finds "1090000000.test.dat" in the working directory, replays it
    and delivers the same callback lines; Stop / Start / Stop works."""
    import subprocess
    from libadsb_amd import build
    iq, _ = synth.fill_range(900, 24)
    path = tmp_path / "1090000000.test.dat"
    with open(path, "wb") as f:
        f.write(iq.tobytes())
        f.write(iq[:7777].tobytes())
    h = A.Handler1090()
    n, fr, ac, nbuf, sec = h.run_replay(str(path))
    ofr, oac = H.oracle_run(iq, BB)
    assert nbuf == 24 and n == len(ofr) > 1000
    ofr = ofr.copy()
    ofr["offset"] %= BB // 2  # every slot is its own HandleData call: frame offsets count from the start of that call
    H.assert_streams_equal(fr, ac, ofr, oac)
    h.close()
    exe = build.build_cxx_test()
    o = O.Oracle1090(sample_clock_hz=0)  # wall clock, like the C++ handler and the reference
    _, oac = H.oracle_run(iq, BB, oracle=o)
    want = [t.encode("latin-1") for t in H.callback_text(oac)]
    out = subprocess.run([exe, "--provider", str(len(want))], capture_output=True, timeout=180, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    got = [l for l in out.stdout.split(b"\n") if l and not l.startswith(b"/opt/amdgpu")]
    assert got == want


def test_random_generator_settings_slice(scanner):
    """A slice of the wide sweep (tools/fuzz_many.py: thousands of settings on an MI355X, result under profiles/): 120 random
    generator settings (noise 0..49, dense to empty bands, weak to saturating amplitudes, any mix of DF17 / DF11 / AP-type, bit
    flips, half-sample offsets), three buffers each: records == the C restatement of the emission contract over the oracle's
    probes, and the handler's callback stream == the oracle's sequential loop."""
    rng = np.random.default_rng(20260104)
    for k in range(120):
        cfg = synth.default_cfg(noise_amp=int(rng.integers(0, 50)), mean_spacing=int(rng.choice([0, 250, 600, 2000, 20000])),
                                amp_lo=int(rng.integers(5, 100)), amp_hi=int(rng.integers(100, 129)), pct_df17=int(rng.integers(0, 60)),
                                pct_df11=int(rng.integers(0, 40)), pct_bitflip=int(rng.integers(0, 100)), pct_halfsample=int(rng.integers(0, 100)))
        first = int(rng.integers(0, 10**6))
        iq, _ = synth.fill_range(first, 3, cfg=cfg)
        try:
            H.assert_records_equal(scanner.scan(iq, BB), O.expected_records(iq, BB, dtype=A.RECORD_DTYPE, nthreads=3))
            h = A.Handler1090()
            fr, ac = h.handle_data(iq, BB)
            h.close()
            ofr, oac = H.oracle_run(iq, BB)
            H.assert_streams_equal(fr, ac, ofr, oac)
        except AssertionError as e:
            raise AssertionError("setting %d first_buffer=%d %s: %s" % (k, first, {f[0]: getattr(cfg, f[0]) for f in cfg._fields_}, e))


def test_one_scanner_inputs_of_changing_size(native_libs):
    """The scan kernels add every chunk's record count into per-256-chunk sums that the ordering pass starts from; the sums live in two
    arrays per slot that swap roles per call, each call's ordering pass zeroing the other one in full.  Large, small, empty and large
    inputs through one scanner must all come out as the oracle's records (a stale sum would misplace every later record)."""
    sc = A.Scanner(0)
    BB = A.REF_BUFFER_BYTES
    for k, nbuf in enumerate((96, 2, 0, 96, 1, 0, 0, 40, 96)):
        if nbuf == 0:
            got = sc.scan(np.zeros(0, dtype=np.uint8), BB)
            assert len(got) == 0
            continue
        iq, _ = synth.fill_range(100 * k, nbuf, nthreads=8)
        H.assert_records_equal(sc.scan(iq, BB), H.expected_records(iq, BB))
    sc.close()


@pytest.mark.parametrize("name", ["embedded", "rtlsdr"])
def test_gpu_handler_prints_the_reference_golden_text(native_libs, name):
    """The stream tests/golden_text.py synthesises from the reference's expected-output file, through the GPU handler's HandleData: the
    callback text must be that file, line for line (see test_reference_golden_text_is_reproduced_line_by_line for what this pins)."""
    import golden_text as G
    iq, want = G.build(name)
    h = A.Handler1090()
    fr, ac = h.handle_data(iq, 0)
    h.close()
    assert H.callback_text(ac) == want
    # and in reference-sized buffers, one HandleData call each (the live path); the generator keeps frames away from buffer ends
    h = A.Handler1090()
    got = []
    for b in range(iq.size // BB):
        _, ac = h.handle_data(iq[b * BB:(b + 1) * BB], BB)
        got += H.callback_text(ac)
    h.close()
    assert got == want


def test_packed_outputs_and_the_frameless_handler(native_libs):
    """set_outputs(OUT_PACKED): the ordering pass writes the packed form only; it equals records + decoded fields merged; a fetch of an
    array that was not produced is refused; a handler told that its listener looks at aircraft only delivers the same snapshots."""
    import torch
    iq, _ = synth.fill_range(9, 64)
    d = torch.from_numpy(iq).cuda()
    st = torch.cuda.current_stream().cuda_stream
    sc = A.Scanner()
    sc.submit(d.data_ptr(), d.numel(), BB, st, 0)
    rec, dec = sc.fetch_decoded(0)
    sc.set_outputs(A.OUT_PACKED)
    sc.submit(d.data_ptr(), d.numel(), BB, st, 0)
    pk = sc.fetch_packed(0)
    assert np.array_equal(pk, A.pack_records(rec, dec)) and len(pk) > 3000
    sc.submit(d.data_ptr(), d.numel(), BB, st, 1)
    with pytest.raises(A.AdsbAmdError):
        sc.fetch(1)
    sc.set_outputs(A.OUT_RECORDS | A.OUT_DECODED | A.OUT_PACKED)
    sc.submit(d.data_ptr(), d.numel(), BB, st, 1)
    rec2, dec2 = sc.fetch_decoded(1)
    assert np.array_equal(rec2, rec) and np.array_equal(dec2, dec)
    sc.close()
    h1, h2 = A.Handler1090(), A.Handler1090()
    h2.set_frames(False)
    fr1, ac1 = h1.handle_data(iq, BB)
    fr2, ac2 = h2.handle_data(iq, BB)
    assert np.array_equal(ac1, ac2) and not fr2["msg"].any() and np.array_equal(fr1["offset"], fr2["offset"]) and np.array_equal(fr1["addr"], fr2["addr"])
    ofr, oac = H.oracle_run(iq, BB)
    H.assert_streams_equal(fr1, ac1, ofr, oac)
    h1.close()
    h2.close()
