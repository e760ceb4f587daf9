"""Pins for the CPU oracle: reference outputs recorded in SURVEY.md Appendix B, table known-answers visible in the
reference text, and the reference's golden text format (tests/golden/reference/*.txt are data files of the
reference's own test suite; their input captures are not in the snapshot, so only their format is usable)."""
import json
import os
import re

import numpy as np
import pytest

from oracle import oracle_py as O

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEC = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_appendix_b.json")))


def modulate(hexmsg, amp=60, i_only=True):
    """Clean 2 samples/bit pulse train for one frame (layout per reference ADSB1090.cpp:749-771)."""
    msg = bytes.fromhex(hexmsg)
    nbits = len(msg) * 8
    lvl = np.zeros((8 + nbits) * 2, dtype=np.int32)
    lvl[[0, 2, 7, 9]] = amp
    for b in range(nbits):
        bit = (msg[b >> 3] >> (7 - (b & 7))) & 1
        lvl[16 + 2 * b + (0 if bit else 1)] = amp
    iq = np.full(2 * lvl.size, 127, dtype=np.uint8)
    iq[0::2] = 127 + lvl
    return iq


def place(buf, iq_frame, at):
    buf[2 * at: 2 * at + iq_frame.size] = iq_frame


def test_survey_recorded_reference_outputs():
    o = O.Oracle1090()
    buf = np.full(262144, 127, dtype=np.uint8)
    at = 1000
    for v in VEC["frames"]:
        place(buf, modulate(v["hex"]), at)
        at += 3000
    fr, ac = o.handle_data(buf)
    assert len(fr) == len(VEC["frames"])
    for v, f, a in zip(VEC["frames"], fr, ac):
        assert bytes(f["msg"]).hex().upper() == v["hex"]
        for k, want in v["expect"].items():
            got = a[k]
            if k == "callsign":
                got = got.decode().ljust(8)
            assert got == want, (v["hex"], k, got, want)


def test_survey_one_bit_repair_case():
    v = VEC["one_bit_repair"]
    o = O.Oracle1090()
    buf = np.full(262144, 127, dtype=np.uint8)
    place(buf, modulate(v["hex"]), 500)
    fr, ac = o.handle_data(buf)
    assert len(fr) == 1 and fr[0]["errorbit"] == v["fixed_bit"]
    assert bytes(fr[0]["msg"]).hex().upper() == v["decodes_as"]
    assert ac[0]["callsign"].decode().ljust(8) == "KLM1023 "


def test_survey_buffer_edge_loss_count():
    # The survey's probe: noise +-3, one DF17 frame every 2000 samples from sample 500 (amplitude 60 + s % 40, low
    # samples I=128,Q=126) over 256 reference buffers -> 16777 injected, 16747 callbacks: the 30 frames that start
    # inside the last 240 samples of a buffer are lost because HandleData keeps no carry-over.
    e = VEC["edge_loss"]
    nbuf, n = e["buffers"], 131072
    rng = np.random.default_rng(12345)
    stream = (127 + rng.integers(-3, 4, size=2 * n * nbuf)).astype(np.uint8)
    msg = bytes.fromhex("8D4840D6202CC371C32CE0576098")
    hi = np.zeros(240, dtype=bool)
    hi[[0, 2, 7, 9]] = True
    for b in range(112):
        bit = (msg[b >> 3] >> (7 - (b & 7))) & 1
        hi[16 + 2 * b + (0 if bit else 1)] = True
    starts = np.arange(500, n * nbuf - 300, e["injected_every"])
    assert len(starts) == e["injected"]
    for s in starts:
        seg = stream[2 * s: 2 * (s + 240)].reshape(240, 2)
        seg[:, 0] = np.where(hi, 127 + 60 + int(s % 40), 128)
        seg[:, 1] = np.where(hi, 127, 126)
    o = O.Oracle1090()
    decoded = 0
    for b in range(nbuf):
        fr, _ = o.handle_data(stream[2 * n * b: 2 * n * (b + 1)])
        decoded += len(fr)
    assert decoded == e["decoded"]


def test_checksum_table_known_entries():
    # values visible in the reference's table text (ADSB1090.cpp:266-275)
    known = {0: 0x3935EA, 1: 0x1C9AF5, 2: 0xF1B77E, 24: 0x939020, 56: 0x018567, 57: 0xFF38B7, 86: 0x001C1B, 87: 0xFFF409, 88: 0, 111: 0}
    for idx, val in known.items():
        assert O.lib().oracle1090_checksum_entry(idx) == val
    # single-bit syndromes are pairwise distinct for both lengths (so "first hit" == "the hit")
    for bits in (112, 56):
        syn = set()
        for j in range(bits):
            m = bytearray(14)
            m[j // 8] ^= 0x80 >> (j % 8)
            nb = bits // 8
            s = O.lib().oracle1090_checksum(bytes(m), bits) ^ int.from_bytes(m[nb - 3:nb], "big")
            assert s != 0 and s not in syn
            syn.add(s)


LIT = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_literals.json")))


def test_every_literal_of_the_reference_checksum_table():
    """All 112 entries the reference spells out (ADSB1090.cpp:266-275), not a sample of them: the oracle's table, which it regenerates from
    the polynomial, and the remainders x^(111 - b) mod (x^24 + 0xFFF409) computed here."""
    tab = LIT["modes_checksum_table"]
    assert len(tab) == 112 and tab[88:] == [0] * 24
    for idx, val in enumerate(tab):
        assert O.lib().oracle1090_checksum_entry(idx) == val, idx
    t = 0xFFF409
    for j in range(87, -1, -1):
        assert tab[j] == t, j
        t <<= 1
        if t & 0x1000000:
            t ^= 0x1FFF409
    # the checksum of a message is the XOR of the entries of its set bits (:277-291): every single-bit message, both lengths
    for bits, off in ((112, 0), (56, 56)):
        for j in range(bits):
            m = bytearray(14)
            m[j // 8] ^= 0x80 >> (j % 8)
            assert O.lib().oracle1090_checksum(bytes(m), bits) == tab[j + off]


def test_every_literal_of_the_reference_nl_table(native_libs):
    """The 58 latitude boundaries of CprNlFunction (ADSB1090.cpp:996-1054) as literals: NL drops by one exactly AT each boundary (the
    comparison is `lat < boundary`), symmetrically about the equator -- in the oracle and in the product's table-driven look-up
    (resolver1090.cpp; exported as adsb_amd_cpr_nl)."""
    import libadsb_amd as A
    bounds = LIT["cpr_nl_boundaries"]
    assert len(bounds) == 58
    for fn in (O.lib().oracle1090_cpr_nl, A.lib().adsb_amd_cpr_nl):
        assert fn(0.0) == 59 and fn(90.0) == 1 and fn(-90.0) == 1
        for k, b in enumerate(bounds):
            below, at = np.nextafter(b, 0.0), b
            for sign in (1.0, -1.0):
                assert fn(sign * below) == 59 - k, (k, b)
                assert fn(sign * at) == 58 - k, (k, b)


def test_the_reference_character_set_of_identification_frames(native_libs):
    """AisCharset (ADSB1090.cpp:608) as a literal: all 64 six-bit codes through an identification frame (DF17, type 4), eight per frame --
    through the oracle's tracker (the call sign it hands the listener) and through the product's host field decoder."""
    import libadsb_amd as A
    cs = LIT["ais_charset"]
    assert len(cs) == 64
    for base in range(0, 64, 8):
        codes = list(range(base, base + 8))
        me = (4 << 51) | sum(c << (42 - 6 * i) for i, c in enumerate(codes))  # type 4, category 0, eight characters
        msg = bytearray([17 << 3 | 5, 0x48, 0x40, 0xD6]) + me.to_bytes(7, "big") + bytes(3)
        crc = O.lib().oracle1090_checksum(bytes(msg), 112)
        msg[11:14] = crc.to_bytes(3, "big")
        want = cs[base:base + 8]
        buf = np.full(A.REF_BUFFER_BYTES, 127, dtype=np.uint8)
        place(buf, modulate(bytes(msg).hex()), 1000)
        o = O.Oracle1090()
        frames, aircraft = o.handle_data(buf)
        assert len(frames) == 1 and aircraft[0]["callsign"] == want.encode("latin-1"), (base, aircraft)
        rec = np.zeros(1, dtype=A.RECORD_DTYPE)
        rec["df"], rec["nbits"], rec["errorbit"] = 17, 112, -1
        rec["msg"][0] = np.frombuffer(bytes(msg), dtype=np.uint8)
        d = A.decode_records_host(rec)[0]
        got = int(d["a"]).to_bytes(4, "little") + int(d["b"]).to_bytes(4, "little")
        assert got == want.encode("latin-1"), (base, got)


def test_magnitude_lut_properties():
    lut = np.ctypeslib.as_array(O.lib().oracle1090_mag_lut(), shape=(129 * 129,)).reshape(129, 129)
    assert lut[0, 0] == 0 and lut[1, 0] == 360 and lut[3, 4] == 1800 and lut[128, 128] == 65167
    s = (np.arange(129)[:, None] ** 2 + np.arange(129)[None, :] ** 2).ravel()
    m = lut.ravel().astype(np.int64)
    # exact integer characterisation used by the product's table builder: m*m - m < 129600*s <= m*m + m
    nz = s > 0
    assert np.all(m[nz] * m[nz] - m[nz] < 129600 * s[nz]) and np.all(129600 * s[nz] <= m[nz] * m[nz] + m[nz])
    # strictly increasing in s: comparisons between magnitudes equal comparisons between s (stage 1 on the GPU)
    order = np.argsort(s, kind="stable")
    ds, dm = np.diff(s[order]), np.diff(m[order])
    assert np.all(dm[ds > 0] > 0) and np.all(dm[ds == 0] == 0)


def test_callback_text_format_matches_reference_goldens():
    pat = re.compile(rb"^([0-9a-f]+)\[(.{8})\]: Pos=([+-]\d+\.\d\d):([+-]\d+\.\d\d)\^(-?\d{5}) Speed=(\d{3,}) Count=(\d+)$", re.S)
    import gzip
    # (the two 978 files come out of the same listener -- tests/test_1090.cpp:13-42 formats every callback, whichever handler made it --, so
    # their 42 + 9053 lines pin the same formatter; the larger one is kept compressed)
    for name in ("TestEmbedded_modes1.bin.txt", "TestEnv_rtlsdr_10902021-06-25-07-39-00.txt", "TestEnv_rtlsdr_978_2021-06-16-20-21-39.txt",
                 "TestEnv_rtlsdr_978_2021-06-16-22-30-43.txt.gz"):
        path = os.path.join(ROOT, "tests", "golden", "reference", name)
        lines = (gzip.open(path, "rb") if name.endswith(".gz") else open(path, "rb")).read().split(b"\n")
        lines = [l for l in lines if l]
        assert len(lines) in (260, 78, 42, 9053)
        for l in lines:
            m = pat.match(l)
            assert m, l
            addr, cs, lat, lon, alt, spd, cnt = m.groups()
            # hundredths of a degree survive the round trip through lat1e7 -> "%+03.2f"
            lat1e7 = int(round(float(lat) * 100)) * 100000
            lon1e7 = int(round(float(lon) * 100)) * 100000
            text = O.format_aircraft(int(addr, 16), cs, lat1e7, lon1e7, int(alt), int(spd), int(cnt))
            assert text.encode("latin-1") == l


def test_uat_phase_lut_known_points():
    lut = np.empty(65536, dtype=np.uint16)
    O.lib().oracle978_phase_lut(lut.ctypes.data)
    # atan2 quadrant anchors of UAT978.cpp:76-100: phase = round(32768 * (atan2(Q-127.5, I-127.5) + pi) / pi)
    assert lut[255 | (128 << 8)] == int(round(32768 * (np.arctan2(0.5, 127.5) + np.pi) / np.pi)) % 65536 or lut[255 | (128 << 8)] == 65535
    assert lut[0 | (0 << 8)] == int(round(32768 * (np.arctan2(-127.5, -127.5) + np.pi) / np.pi))
    assert lut[128 | (255 << 8)] == int(round(32768 * (np.arctan2(127.5, 0.5) + np.pi) / np.pi))


def test_synth_is_deterministic():
    import hashlib
    from libadsb_amd import synth
    a = synth.fill(0)
    b = synth.fill(0)
    assert np.array_equal(a, b)
    digest = hashlib.sha256(a.tobytes()).hexdigest()
    want = open(os.path.join(ROOT, "tests", "golden", "synth_buffer0.sha256")).read().strip()
    assert digest == want


PUBLIC_FRAMES = [  # extended squitters printed in public ADS-B tutorials (e.g. "The 1090 Megahertz Riddle"): independent of the survey
    "8D4840D6202CC371C32CE0576098", "8D40621D58C382D690C8AC2863A7", "8D40621D58C386435CC412692AD6", "8D485020994409940838175B284F",
    "8DA05F219B06B6AF189400CBC33F", "8D406B902015A678D4D220AA4BDA", "8D40058B58C901375147EFD09357", "8D40058B58C904A87F402D3B8C59",
]


def test_parity_of_publicly_documented_frames():
    """The checksum restated from ADSB1090.cpp:266-291 equals the transmitted parity of frames published elsewhere."""
    for h in PUBLIC_FRAMES:
        m = bytes.fromhex(h)
        assert O.lib().oracle1090_checksum(m, 112) == int.from_bytes(m[11:14], "big"), h


def test_public_frames_decode_through_the_oracle_and_one_flipped_bit_is_repaired():
    o = O.Oracle1090()
    buf = np.full(262144, 127, dtype=np.uint8)
    for k, h in enumerate(PUBLIC_FRAMES):
        place(buf, modulate(h), 500 + 3000 * k)
    damaged = bytearray(bytes.fromhex("8D406B902015A678D4D220AA4BDA"))
    damaged[6] ^= 0x04  # message bit 53
    place(buf, modulate(damaged.hex()), 500 + 3000 * len(PUBLIC_FRAMES))
    fr, ac = o.handle_data(buf)
    assert [bytes(f["msg"]).hex().upper() for f in fr] == PUBLIC_FRAMES + ["8D406B902015A678D4D220AA4BDA"]
    assert [int(f["errorbit"]) for f in fr] == [-1] * len(PUBLIC_FRAMES) + [53]
    assert ac[0]["callsign"].decode().ljust(8) == "KLM1023 " and ac[5]["callsign"].decode().ljust(8) == "EZY85MH "
    # the classic CPR pair (even then odd, odd is the newer one): 52.26578 N, 3.93891 E
    assert abs(ac[2]["lat1e7"] - 522657800) < 20 and abs(ac[2]["lon1e7"] - 39389100) < 50 and ac[2]["altitude"] == 38000


def test_float_magnitude_estimate_stays_within_the_bound_the_kernel_assumes():
    """scan1090.hip decides stage 2 and parts of the demodulation on e(s) = 360 * sqrtf(s) (s saturated at 32767) and trusts
    |e(s) - m(s)| < kEstErr = 1.6 for the reference magnitude m(s).  Over every reachable s, with a sqrt that is off by one
    ulp either way, the error stays below 1.05."""
    vals = np.array(sorted({i * i + q * q for i in range(129) for q in range(129)}), dtype=np.int64)
    m = np.array([O.lib().oracle1090_magnitude_of(int(x)) for x in vals]) if hasattr(O.lib(), "oracle1090_magnitude_of") else \
        np.floor(np.sqrt(vals.astype(np.float64)) * 360 + 0.5)
    root = np.sqrt(np.minimum(vals, 32767).astype(np.float32))
    worst = 0.0
    for r in (root, np.nextafter(root, np.float32(np.inf)), np.nextafter(root, np.float32(-np.inf))):
        worst = max(worst, float(np.abs(np.float32(360.0) * r - m).max()))
    assert worst < 1.05


@pytest.mark.parametrize("name", ["embedded", "rtlsdr"])
def test_reference_golden_text_is_reproduced_line_by_line(name):
    """tests/golden_text.py encodes one Mode S frame per line of the reference's expected-output files (its own encoder) and modulates
    them into a u8 IQ stream; the oracle, and the product's host half fed with the records the GPU contract defines, must print the
    file exactly: which fields a callback changes, altitude before any position, `Pos` at +0.00:+0.00 until a pair decodes, the
    `Speed=%03d` / `^%05d` widths, the NUL call sign of a fresh aircraft, `Count=0`.  The inputs are chosen, not given: this ties the
    formatter, the sticky aircraft state of ADSB1090.cpp:1124-1175 and the field decoders to reference-held text, it is not a parity
    proof against a reference capture (tests/test_1090.cpp:13-42 is the flow it mirrors)."""
    import golden_text as G
    import libadsb_amd as A
    iq, want = G.build(name)
    fr, ac = H.oracle_run(iq, 0)  # the whole stream through one HandleData call, as TestEmbedded does (tests/test_1090.cpp:66-67)
    assert H.callback_text(ac) == want
    rec = H.expected_records(iq, 0)
    n, fr2, ac2 = A.Resolver().feed(rec, iq.size // 2, 1)
    assert n == len(want) and H.callback_text(ac2) == want
    # and in reference-sized buffers, one HandleData call each (the live path): the same text
    o = O.Oracle1090()
    got = []
    for b in range(iq.size // 262144):
        _, ac3 = o.handle_data(iq[b * 262144:(b + 1) * 262144])
        got += H.callback_text(ac3)
    assert got == want
