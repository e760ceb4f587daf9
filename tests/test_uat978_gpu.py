"""UAT 978 on the GPU against oracle/oracle978.c, through the C ABI (parity unpinned: the oracle restates the published
dump978 legacy algorithm, see its header).  Bit-exact: same frames, same payload bytes, same rs_errors, same stream
sample index, same consumed counts."""
import ctypes as C

import numpy as np
import pytest

import libadsb_amd as A
from libadsb_amd import synth
from oracle import oracle_py as O

import uat_helpers as U
from test_uat978_cpu import stale_register_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uat(native_libs):
    u = A.Uat978()
    yield u
    u.close()


def test_phase_lut_is_the_reference_table(uat):
    assert np.array_equal(uat.phase_lut(), O.phase_lut978())


@pytest.mark.parametrize("stream,cfg_over", [
    (0, {}),
    (1, {"noise_amp": 6, "pct_corrupt": 60, "max_bad_bytes": 8}),     # beyond RS capacity for some frames
    (2, {"pct_uplink": 50, "mean_gap_bits": 300}),
    (3, {"amp_lo": 6, "amp_hi": 14, "noise_amp": 4}),                  # weak signals: sync errors, wrong slicing
    (4, {"mean_gap_bits": 40, "pct_uplink": 0}),                       # back-to-back downlink frames
])
@pytest.mark.parametrize("carry_full", [False, True])
def test_handle_data_matches_oracle(native_libs, stream, cfg_over, carry_full):
    cfg = synth.default_cfg978(**cfg_over)
    iq = synth.fill978(stream, 6 * 262144, cfg)
    u, o = A.Uat978(carry_full=carry_full), O.Oracle978(carry_full=carry_full)
    total = 0
    for k in range(6):  # production-sized calls, state carried across them
        part = iq[k * 262144:(k + 1) * 262144]
        got, want = u.handle_data(part), o.handle_data(part)
        assert got == want, (k, len(got), len(want))
        assert u.stream_state() == o.stream_state()
        total += len(want)
    assert total > 20
    u.close()


def test_handle_data_ragged_call_sizes(native_libs):
    iq = synth.fill978(7, 5 * 262144, synth.default_cfg978())
    u, o = A.Uat978(), O.Oracle978()
    rng = np.random.default_rng(7)
    pos = 0
    while pos < iq.size:
        n = int(rng.choice([2, 20000, 65536, 131072, 262144, 300002, 17880]))
        part = iq[pos:pos + n]
        assert u.handle_data(part) == o.handle_data(part), pos
        assert u.stream_state() == o.stream_state()
        pos += n
    u.close()


def test_process_iq_one_long_buffer(uat):
    """process_buffer semantics over 8 Mi samples in one call (the batch path) == the oracle on the LUT-mapped phases."""
    cfg = synth.default_cfg978(pct_corrupt=30, max_bad_bytes=7)
    iq = synth.fill978(11, 16 * 1024 * 1024, cfg)
    phi = O.phase_lut978()[iq.view(np.uint16)]
    want, want_done = O.process_buffer978(phi, offset=12345)
    got, got_done = uat.process_iq(iq, offset=12345)
    assert len(want) > 1000
    assert got == want and got_done == want_done


def test_process_phases_and_short_buffers(uat):
    iq = synth.fill978(12, 2 * 65536, synth.default_cfg978())
    phi = O.phase_lut978()[iq.view(np.uint16)]
    for n in (0, 1, 2, 35, 36, 37, 8903, 8904, 8905, 8906, 8940, 9000, 20001, 65536):
        want = O.process_buffer978(phi[:n])
        got = uat.process_phases(phi[:n])
        assert got == want, n


def test_noise_only_and_constant_input(uat):
    rng = np.random.default_rng(3)
    for iq in (rng.integers(0, 256, 4 * 1024 * 1024, dtype=np.uint8), np.full(1 << 20, 127, dtype=np.uint8),
               np.zeros(1 << 20, dtype=np.uint8)):
        phi = O.phase_lut978()[iq.view(np.uint16)]
        assert uat.process_iq(iq) == O.process_buffer978(phi)


def test_stale_register_detection_needs_an_extra_device_lookup(uat):
    phi, p1, p2 = stale_register_case()
    before = uat.timing()["extra_lookups"]
    got = uat.process_phases(phi)
    assert got == O.process_buffer978(phi)
    assert [f[1] for f in got[0]] == [p1, p2]
    assert uat.timing()["extra_lookups"] == before + 1


def test_adsb_word_on_one_alignment_hides_uplink_word_on_the_other(uat):
    """`else if`: when either register holds the ADS-B check word the uplink word is not looked at on that bit.  Built so
    that the even alignment shows the uplink word and the odd alignment the ADS-B word on the same bit."""
    rng = np.random.default_rng(9)
    a = U.bits_of(U.ADSB_SYNC, 36)
    b = U.bits_of(U.UPLINK_SYNC, 36)
    # sample-level signs: even samples carry word b, odd samples word a (only the first 18 bits matter for the search)
    signs = np.empty(72, dtype=np.int64)
    signs[0::2] = b
    signs[1::2] = a
    d = np.concatenate([np.where(np.array(U.quiet_bits(rng, 80)).repeat(2) > 0, U.STEP, -U.STEP),
                        np.where(signs > 0, U.STEP, -U.STEP),
                        np.where(np.array(U.quiet_bits(rng, 2 * 4600)).repeat(2) > 0, U.STEP, -U.STEP)])
    phi = ((1000 + np.concatenate([[0], np.cumsum(d)])) & 0xFFFF).astype(np.uint16)
    assert uat.process_phases(phi) == O.process_buffer978(phi)


def test_reference_seam_process_buffer(native_libs):
    """init_fec / process_buffer / dump_raw_message: the names UAT978.cpp:9-10 binds."""
    L = A.lib()
    frames = []

    def _dump(updown, data, n, rs):
        frames.append((updown.decode(), bytes(data[:n]), int(rs)))
    cb = A.DUMP_RAW_MESSAGE(_dump)
    L.adsb_amd_uat_set_dump_raw_message(cb)
    try:
        L.init_fec()
        iq = synth.fill978(5, 131072, synth.default_cfg978())
        phi = np.ascontiguousarray(O.phase_lut978()[iq.view(np.uint16)])
        done = L.process_buffer(phi.ctypes.data, phi.size, 777)
        want, want_done = O.process_buffer978(phi, 777)
        assert done == want_done and frames == [f[:3] for f in want] and len(want) > 3
    finally:
        L.adsb_amd_uat_set_dump_raw_message(None)


@pytest.mark.parametrize("over", [
    {"amp_lo": 110, "amp_hi": 127, "noise_amp": 30},                      # clipping at 0 and 255 on most frames
    {"amp_lo": 126, "amp_hi": 127, "noise_amp": 60, "pct_uplink": 40},    # nearly every sample of a frame clipped somewhere on the way round
    {"amp_lo": 2, "amp_hi": 5, "noise_amp": 1, "mean_gap_bits": 200},     # everything within a few LSB of the centre (127 / 128)
])
def test_iq_path_at_full_scale_and_at_the_centre(uat, over):
    """The batch path looks its phases up in one quadrant of the table and derives the rest (uat978.hip, lut2_folded): the coordinates 0 and
    255 (the fold's far edge) and 127 / 128 (its seam) are where a wrong fold would show.  Frames == the oracle's on the whole table."""
    cfg = synth.default_cfg978(**over)
    iq = synth.fill978(77, 16 * 1024 * 1024, cfg)
    b = iq.reshape(-1, 2)
    if over["amp_hi"] >= 127:
        assert int((b == 0).sum()) > 1000 and int((b == 255).sum()) > 1000
    want = O.process_buffer978(O.phase_lut978()[iq.view(np.uint16)])
    assert uat.process_iq(iq) == want
    assert len(want[0]) > 100


def test_device_resident_input(uat):
    import torch
    iq = synth.fill978(13, 8 * 1024 * 1024, synth.default_cfg978())
    d = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    got = uat.process_device(d.data_ptr(), iq.size // 2)
    assert got == O.process_buffer978(O.phase_lut978()[iq.view(np.uint16)])
    t = uat.timing()
    assert t["scan_ms"] > 0


def test_calls_in_flight_equal_the_serial_calls(native_libs):
    """submit / collect: the GPU half of call k + 1 runs while the scan loop of call k walks its records; frames and consumed
    counts call by call as process_iq gives them (and as the oracle does), a fourth submit is refused."""
    import torch
    u = A.Uat978()
    streams = [synth.fill978(40 + k, (4 + 2 * k) * 1024 * 1024, synth.default_cfg978(**over))
               for k, over in enumerate(({}, {"pct_uplink": 40}, {"mean_gap_bits": 40, "pct_corrupt": 60}, {}, {"noise_amp": 6}))]
    dev = [torch.from_numpy(x).cuda() for x in streams]
    torch.cuda.synchronize()
    want = [u.process_device(d.data_ptr(), x.size // 2, offset=1000 * k) for k, (d, x) in enumerate(zip(dev, streams))]
    assert want[0] == O.process_buffer978(O.phase_lut978()[streams[0].view(np.uint16)])
    got = []
    ahead = u.max_in_flight() - 1
    assert 2 <= ahead < len(streams)
    for k in range(ahead):
        u.submit_device(dev[k].data_ptr(), streams[k].size // 2, 1000 * k)
    for k in range(len(streams)):
        if k + ahead < len(streams):
            u.submit_device(dev[k + ahead].data_ptr(), streams[k + ahead].size // 2, 1000 * (k + ahead))
            if k == 0:
                with pytest.raises(A.AdsbAmdError):  # every buffer set of the handle is in use: no further submit
                    u.submit_device(dev[3].data_ptr(), streams[3].size // 2, 0)
                with pytest.raises(A.AdsbAmdError):  # nor a synchronous call on the same handle: it would share side 0's buffers
                    u.process_device(dev[0].data_ptr(), streams[0].size // 2)
                with pytest.raises(A.AdsbAmdError):
                    u.handle_data(streams[0][:262144])
        got.append(u.collect())
    assert got == want
    with pytest.raises(A.AdsbAmdError):
        u.collect()
    assert u.process_device(dev[1].data_ptr(), streams[1].size // 2, offset=1000) == want[1]  # the plain call still works afterwards
    u.close()


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_device_reed_solomon_matches_oracle_including_beyond_capacity(uat, kind):
    """The wave-wide decoder inside the demod kernel, on its own: clean, correctable, uncorrectable and random words."""
    k, nr = {0: (18, 12), 1: (34, 14), 2: (72, 20)}[kind]
    rng = np.random.default_rng(4242 + kind)
    words = []
    for trial in range(6000):
        data = rng.integers(0, 256, k, dtype=np.uint8).tobytes()
        cw = bytearray(data + O.rs_parity978(kind, data))
        mode = trial % 6
        if mode == 5:
            cw = bytearray(rng.integers(0, 256, k + nr, dtype=np.uint8).tobytes())
        else:
            nerr = trial % (nr // 2 + 1) if mode == 0 else int(rng.integers(0, nr // 2 + 5))
            for p in rng.choice(k + nr, nerr, replace=False):
                cw[p] ^= int(rng.integers(1, 256))
        words.append(bytes(cw))
    arr = np.frombuffer(b"".join(words), dtype=np.uint8).reshape(len(words), k + nr)
    res, fixed = uat.rs_decode_device(kind, arr)
    for i, wd in enumerate(words):
        n, out = O.rs_decode978(kind, wd)
        assert (int(res[i]), fixed[i].tobytes()) == (n, out), i


def test_process_iq_ragged_lengths_around_chunk_and_row_edges(uat):
    """The fused kernel works in 32 768-sample chunks, 2 048-sample wave spans, 512-sample rows and a 64-sample halo; the
    stream may end anywhere relative to those.  A frame is placed so that it ends just before the end of the stream."""
    base = synth.fill978(21, 2 * 262144, synth.default_cfg978(mean_gap_bits=200))
    lut = O.phase_lut978()
    sizes = [0, 1, 2, 35, 36, 37, 511, 512, 513, 2047, 2048, 2049, 8903, 8906, 32767, 32768, 32769, 32768 + 63, 32768 + 64, 32768 + 65,
             34815, 34816, 34817, 65535, 65536, 65537, 65536 + 2048 + 1, 98304, 98305, 131071, 200001, 262144]
    for n in sizes:
        iq = base[:2 * n]
        want = O.process_buffer978(lut[iq.view(np.uint16)])
        got = uat.process_iq(iq)
        assert got == want, n
    assert any(len(O.process_buffer978(lut[base[:2 * n].view(np.uint16)])[0]) > 0 for n in sizes)


def test_cxx_drop_in_uat_handler(native_libs, tmp_path):
    """ADSB::test::TryCreateUAT978Handler -> HandleData -> the host's dump_raw_message (defined by the test binary), with the
    thread-local traffic manager published as UAT978.cpp:46 does.  Lines == oracle frames, in order."""
    import subprocess
    from libadsb_amd import build
    build.build_cxx_test()
    iq = synth.fill978(31, 4 * 262144, synth.default_cfg978())
    path = tmp_path / "978.iq"
    iq.tofile(path)
    out = subprocess.run([build.CXX_TEST_978, str(path), "262144"], capture_output=True, timeout=120)
    assert out.returncode == 0, out.stderr.decode()
    o = O.Oracle978()
    want = []
    for k in range(4):
        want += ["%s %d %d %s" % (ud, len(p), rs, p.hex()) for ud, p, rs, _ in o.handle_data(iq[k * 262144:(k + 1) * 262144])]
    assert out.stdout.decode().splitlines() == want and len(want) > 10


def test_full_size_1gib_stream_equals_oracle(uat):
    """BASELINE configs[4] at full size: 512 Mi samples in one process_buffer, device resident, every frame compared."""
    import torch
    piece = 64 << 20
    dev = torch.empty(16 * piece, dtype=torch.uint8, device="cuda")
    lut = O.phase_lut978()
    phi = np.empty(8 * piece, dtype=np.uint16)
    for k in range(16):
        h = synth.fill978(100 + k, piece, synth.default_cfg978())
        dev[k * piece:(k + 1) * piece].copy_(torch.from_numpy(h))
        phi[k * (piece // 2):(k + 1) * (piece // 2)] = lut[h.view(np.uint16)]
    torch.cuda.synchronize()
    got = uat.process_device(dev.data_ptr(), dev.numel() // 2)
    want = O.process_buffer978(phi)
    assert len(want[0]) > 50000
    assert got == want


def test_decisions_on_the_device_equal_the_host_scan_loop(native_libs):
    """Which frames the scan loop takes is decided on the device (successor function + pointer jumping over blocks of 4 096 matches);
    set_host_loop(True) walks the reference's loop on the host over the same records, as rounds 1-2 did.  Same frames, same consumed
    counts, same number of positions reached through stale register bits -- on a stream with more matches than two blocks, on the
    stale-register case, and on streams too short for a frame."""
    cfg = synth.default_cfg978(pct_corrupt=30, max_bad_bytes=7)
    iq = synth.fill978(21, 128 << 20, cfg)
    phi_stale, _, _ = stale_register_case()
    u = A.Uat978()
    on_device = u.process_iq(iq, offset=77)
    matches = u.timing()["candidates"]
    extras_device = u.timing()["extra_lookups"]
    stale_device = u.process_phases(phi_stale)
    short_device = [u.process_iq(iq[:n]) for n in (0, 2, 72, 8904 * 2, 8906 * 2, 9100 * 2)]
    u.set_host_loop(True)
    before = u.timing()["extra_lookups"]
    on_host = u.process_iq(iq, offset=77)
    extras_host = u.timing()["extra_lookups"] - before
    assert u.process_phases(phi_stale) == stale_device
    assert [u.process_iq(iq[:n]) for n in (0, 2, 72, 8904 * 2, 8906 * 2, 9100 * 2)] == short_device
    u.close()
    assert matches > 2 * 4096 and len(on_device[0]) > 2 * 4096
    assert on_device == on_host
    assert extras_device == extras_host


def test_more_stale_register_frames_than_the_device_keeps_fall_back_to_the_host_loop(native_libs):
    """The device keeps up to 4 096 frames per call that the loop reaches through stale register bits; a call with more is walked by
    the host loop.  With the capacity lowered to zero every such frame overflows: same frames as the oracle, and the host's
    look-ups are counted."""
    phi, p1, p2 = stale_register_case()
    cfg = synth.default_cfg978(pct_corrupt=30, max_bad_bytes=7)
    u = A.Uat978()
    want_stale = u.process_phases(phi)
    for seed in range(31, 41):  # a generated stream with such frames of its own (about 19 per GiB): the first of these seeds that has one
        iq = synth.fill978(seed, 128 << 20, cfg)
        before = u.timing()["extra_lookups"]
        want_long = u.process_iq(iq)
        if u.timing()["extra_lookups"] > before:
            break
    else:
        raise AssertionError("no generated stream with a frame behind stale register bits")
    u.set_extra_capacity(0)
    before = u.timing()["extra_lookups"]
    assert u.process_phases(phi) == want_stale == O.process_buffer978(phi)
    assert u.timing()["extra_lookups"] == before + 1
    assert u.process_iq(iq) == want_long
    u.set_extra_capacity(4096)
    assert u.process_iq(iq) == want_long
    u.close()


@pytest.mark.parametrize("world", [2, 3, 7])
def test_a_stream_cut_into_parts_gives_the_frames_of_the_whole(native_libs, world):
    """SURVEY section 8e, "shard with a halo": one process_buffer over a stream cut over `world` handles (adsb_amd_uat_part_scan /
    _finish: every part's search and demodulation independent of the others, only the bit at which the scan loop stands travels from
    part to part) == the same stream in one call: frames, payloads, rs_errors, sample indices, consumed count.  Streams dense enough that
    every cut lands inside or right behind a frame, uplink frames included."""
    import torch
    from libadsb_amd import shard
    handles = [A.Uat978() for _ in range(world)]
    whole = A.Uat978()
    for seed, over in ((51, {}), (52, {"mean_gap_bits": 40, "pct_uplink": 30}), (53, {"mean_gap_bits": 200, "pct_corrupt": 60, "max_bad_bytes": 7})):
        iq = synth.fill978(seed, 24 << 20, synth.default_cfg978(**over))
        dev = torch.from_numpy(iq).cuda()
        torch.cuda.synchronize()
        want = whole.process_device(dev.data_ptr(), iq.size // 2, offset=5000)
        got = shard.uat_run_parts(handles, dev.data_ptr(), iq.size // 2, offset=5000)
        assert len(want[0]) > 1000
        assert got == want, (seed, world)
    # a part that was scanned and not finished is dropped by any other call on its handle
    handles[0].part_scan(dev.data_ptr(), 1 << 20)
    assert handles[0].process_device(dev.data_ptr(), 1 << 20) == whole.process_device(dev.data_ptr(), 1 << 20)
    with pytest.raises(A.AdsbAmdError):
        handles[0].part_finish(0, 1 << 20, 0, True)
    # a stream too short for `world` parts: the first part whose tail reaches the end takes the rest
    short = synth.fill978(54, 1 << 20, synth.default_cfg978())
    dev = torch.from_numpy(short).cuda()
    torch.cuda.synchronize()
    assert shard.uat_run_parts(handles, dev.data_ptr(), short.size // 2) == whole.process_device(dev.data_ptr(), short.size // 2)
    for u in handles + [whole]:
        u.close()


def test_bench_cuts_one_uat_stream_over_two_ranks_rehearsal(native_libs):
    """`bench.py --workload uat978 --gpus 2`, rehearsed with two ranks on this one GPU (gloo): besides the replicas figure the line carries
    the one-stream job -- each rank its part of a 2 x 64 MiB stream, the loop's position passed from rank 0 to rank 1 --, whose frame
    count and consumed count must be those of one call over the whole stream."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "uat978", "--gpus", "2", "--rehearse-on-one-gpu", "--mib", "64",
                          "--steps", "3", "--warmup", "1", "--cpu-buffers", "0"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    one = line["one_stream_cut_over_ranks"]
    assert one["equals_one_call"] is True and one["frames"] > 5000 and one["samples_in_the_stream"] == 2 * (64 << 20) // 2


@pytest.mark.parametrize("seed", range(6))
def test_fuzzed_phase_streams(uat, seed):
    """Random phase streams salted with sync words at random places and alignments, whole and truncated, clean and with wrong
    bits, followed by random or valid (and randomly corrupted) frames: exercises false matches, both alignments, the
    next-sample variant, the else-if rule, failed sync re-checks, Reed-Solomon successes and failures, jumps and the
    stale-register window right after them."""
    rng = np.random.default_rng(9780 + seed)
    step = np.zeros(400000, dtype=np.int64)
    step[:] = np.where(rng.integers(0, 2, step.size) > 0, 1, -1) * rng.integers(200, 9000, step.size)
    pos = 3000
    while pos < step.size - 12000:
        kind = rng.integers(0, 3)          # 0 short, 1 long, 2 uplink
        word = U.UPLINK_SYNC if kind == 2 else U.ADSB_SYNC
        bits = U.bits_of(word, 36)
        if rng.random() < 0.3:
            bits = bits[int(rng.integers(1, 4)):]                      # sync word missing its first bits
        for k in rng.choice(len(bits), int(rng.integers(0, 7)) if rng.random() < 0.4 else 0, replace=False):
            bits[k] ^= 1                                                # wrong sync bits (up to 6: beyond the tolerance too)
        if kind == 0:
            frame = U.short_frame(bytes([0x00]) + rng.integers(0, 256, 17, dtype=np.uint8).tobytes())
        elif kind == 1:
            frame = U.long_frame(bytes([int(rng.integers(1, 32)) << 3]) + rng.integers(0, 256, 33, dtype=np.uint8).tobytes())
        else:
            frame = U.uplink_frame(rng.integers(0, 256, 432, dtype=np.uint8).tobytes())
        frame = bytearray(frame)
        r = rng.random()
        if r < 0.25:
            frame = bytearray(rng.integers(0, 256, len(frame), dtype=np.uint8).tobytes())   # not a code word at all
        elif r < 0.7:
            for k in rng.choice(len(frame), int(rng.integers(1, 14)), replace=False):
                frame[k] ^= int(rng.integers(1, 256))
        bits = bits + U.bytes_to_bits(frame)
        amp = int(rng.integers(300, 9000))
        d = np.repeat(np.where(np.array(bits) > 0, amp, -amp), 2) + rng.integers(-amp // 3, amp // 3 + 1, 2 * len(bits))
        start = pos + int(rng.integers(0, 2))                                               # either sample alignment
        step[start:start + d.size] = d
        pos = start + d.size + (0 if rng.random() < 0.3 else int(rng.integers(0, 3000)))      # some frames back to back
    phi = ((np.cumsum(step) + 12345) & 0xFFFF).astype(np.uint16)
    want = O.process_buffer978(phi)
    assert uat.process_phases(phi) == want
    assert len(want[0]) > 5
    cut = int(rng.integers(20000, 390000))
    assert uat.process_phases(phi[:cut]) == O.process_buffer978(phi[:cut])


def test_1090_and_978_handlers_run_concurrently_on_one_gpu(native_libs):
    """libadsb runs the two handlers on two consumer threads (RTLSDR.hpp:470-473).  Each has its own context and streams;
    interleaved on one GPU they must produce what they produce alone."""
    import threading
    import helpers as H
    BB = A.REF_BUFFER_BYTES
    iq1090, _ = synth.fill_range(70, 24)
    iq978 = synth.fill978(41, 24 * BB, synth.default_cfg978())
    want1090 = H.oracle_run(iq1090, BB)
    o = O.Oracle978()
    want978 = []
    for k in range(24):
        want978 += o.handle_data(iq978[k * BB:(k + 1) * BB])
    got = {}

    def run1090():
        h = A.Handler1090()
        frames, aircraft = [], []
        for k in range(24):
            fr, ac = h.handle_data(iq1090[k * BB:(k + 1) * BB], BB)
            fr = fr.copy()
            fr["offset"] += k * (BB // 2)  # frame offsets are relative to the call; the oracle helper numbers them through
            frames.append(fr), aircraft.append(ac)
        got["1090"] = (np.concatenate(frames), np.concatenate(aircraft))
        h.close()

    def run978():
        u = A.Uat978()
        out = []
        for k in range(24):
            out += u.handle_data(iq978[k * BB:(k + 1) * BB])
        got["978"] = out
        u.close()

    threads = [threading.Thread(target=run1090), threading.Thread(target=run978)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    H.assert_streams_equal(got["1090"][0], got["1090"][1], want1090[0], want1090[1])
    assert got["978"] == want978 and len(want978) > 50


def test_handle_data_one_large_call(native_libs):
    """64 MiB through HandleData in one call: 500+ staging rounds of the reference's loop (UAT978.cpp:50-59) on the device."""
    iq = synth.fill978(55, 64 << 20, synth.default_cfg978())
    for full in (False, True):
        u, o = A.Uat978(carry_full=full), O.Oracle978(carry_full=full)
        got, want = u.handle_data(iq), o.handle_data(iq)
        assert got == want and len(want) > 3000
        assert u.stream_state() == o.stream_state()
        u.close()


def test_random_generator_settings_and_call_sizes_slice(native_libs):
    """A slice of the wide sweep (tools/fuzz_many.py): 60 random UAT generator settings, the stream cut into random HandleData
    calls, both carry modes: frames and stream state == the oracle after every call."""
    rng = np.random.default_rng(9780104)
    BBL = 262144
    for k in range(60):
        cfg = synth.default_cfg978(noise_amp=int(rng.integers(0, 30)), amp_lo=int(rng.integers(3, 40)), amp_hi=int(rng.integers(40, 120)),
                                   mean_gap_bits=int(rng.choice([0, 30, 300, 3000, 50000])), pct_uplink=int(rng.integers(0, 100)),
                                   pct_long=int(rng.integers(0, 100)), pct_corrupt=int(rng.integers(0, 100)), max_bad_bytes=int(rng.integers(1, 16)))
        iq = synth.fill978(int(rng.integers(0, 10**6)), 3 * BBL, cfg)
        full = bool(rng.integers(0, 2))
        uu, oo = A.Uat978(carry_full=full), O.Oracle978(carry_full=full)
        pos = 0
        while pos < iq.size:
            n = int(rng.choice([BBL, BBL, 2 * BBL, 65536, 30000, 100002]))
            part = iq[pos:pos + n]
            assert uu.handle_data(part) == oo.handle_data(part), (k, pos, {f[0]: getattr(cfg, f[0]) for f in cfg._fields_})
            assert uu.stream_state() == oo.stream_state()
            pos += n
        uu.close()
