"""Shared test helpers: the expected GPU record set derived from the oracle's state-free probes.

The emission rules restated here are the contract of include/adsb_amd.h (adsb_amd_record_t):
one record per (offset, pass) that the reference could accept.  They are written against the
oracle's probe fields so that a GPU record array can be compared element by element.
"""
import numpy as np

import libadsb_amd as A
from oracle import oracle_py as O

AP_DFS = (0, 4, 5, 16, 20, 21, 24)


def _rec(buffer, j, msg, nbits, errorbit, df, flags, addr, delta):
    r = np.zeros(1, dtype=A.RECORD_DTYPE)[0]
    r["buffer"], r["offset"], r["addr"] = buffer, j, addr
    r["nbits"], r["errorbit"], r["df"], r["flags"] = nbits, errorbit, df, flags
    m = np.frombuffer(bytes(msg), dtype=np.uint8).copy()
    m[nbits // 8:] = 0  # contract: bytes beyond the message length are zero (sliced noise in the reference, never read)
    r["msg"] = m
    return r


def records_from_probe(buffer, j, p):
    out = []
    if p.p_errors[0] != 0:
        assert p.p_errors[1] != 0  # bit 0 is never rescaled, the retry cannot clear it
        return out
    if not p.p_energy_ok[0]:
        return out
    for k in (0, 1):
        if k == 1:
            if not p.phase_applied or not p.p_energy_ok[1] or p.p_errors[1] != 0:
                break
        df, nbits = p.p_df[k], p.p_nbits[k]
        flags = 0 if k == 0 else (A.F_PASS2 | A.F_PHASE)
        if df in (11, 17):
            if p.p_crc_state[k] in (1, 2):
                msg = bytes(p.p_fixed[k])
                addr = (msg[1] << 16) | (msg[2] << 8) | msg[3]
                out.append(_rec(buffer, j, msg, nbits, p.p_errorbit[k], df, flags, addr, p.p_delta[k]))
                break  # accepted without state: the reference never retries
        elif df in AP_DFS:
            out.append(_rec(buffer, j, bytes(p.p_msg[k]), nbits, -1, df, flags | A.F_NEEDS_ICAO, p.p_ap_addr[k], p.p_delta[k]))
    return out


def expected_records(iq, buffer_bytes=0):
    """Expected adsb_amd_record_t array for one scan call over `iq` (uint8)."""
    iq = np.ascontiguousarray(iq, dtype=np.uint8)
    bb = buffer_bytes or (iq.size & ~1)
    nbuf = iq.size // bb if bb else 0
    if buffer_bytes == 0 and iq.size < 480:
        nbuf = 0
    recs = []
    for b in range(nbuf):
        mag = O.magnitude(iq[b * bb:(b + 1) * bb])
        for j in O.gate_offsets(mag):
            recs.extend(records_from_probe(b, int(j), O.probe_at(mag, int(j))))
    return np.array(recs, dtype=A.RECORD_DTYPE) if recs else np.zeros(0, A.RECORD_DTYPE)


def oracle_run(iq, buffer_bytes=0, oracle=None):
    """Reference-order processing: one HandleData call per buffer. Returns (frames, aircraft) with frames['offset']
    rewritten to buffer * samples_per_buffer + j so that it is comparable with the product's frame stream."""
    iq = np.ascontiguousarray(iq, dtype=np.uint8)
    o = oracle or O.Oracle1090()
    bb = buffer_bytes or iq.size
    nbuf = iq.size // bb if bb else 0
    frs, acs = [], []
    for b in range(nbuf):
        fr, ac = o.handle_data(iq[b * bb:(b + 1) * bb])
        fr = fr.copy()
        fr["offset"] += b * (bb // 2)
        fr["msg"][fr["nbits"] == 56, 7:] = 0  # see _rec(): the product zeroes the unused tail of short frames
        frs.append(fr)
        acs.append(ac)
    if not frs:
        return np.zeros(0, O.FRAME_DTYPE), np.zeros(0, O.AIRCRAFT_DTYPE)
    return np.concatenate(frs), np.concatenate(acs)


def assert_records_equal(got, want):
    assert got.dtype == want.dtype
    if got.shape != want.shape or not np.array_equal(got, want):
        n = min(len(got), len(want))
        for i in range(n):
            if got[i] != want[i]:
                raise AssertionError("record %d differs\n got  %r\n want %r (lens %d / %d)" % (i, got[i], want[i], len(got), len(want)))
        raise AssertionError("record count differs: got %d want %d; first extra: %r" %
                             (len(got), len(want), (got[n] if len(got) > n else want[n])))


def assert_streams_equal(fr_a, ac_a, fr_b, ac_b):
    """Frame streams and aircraft snapshot streams must agree field by field, in order."""
    assert len(fr_a) == len(fr_b), "accepted frame count differs: %d vs %d" % (len(fr_a), len(fr_b))
    # the product zeroes the unused tail of short frames (the reference keeps sliced noise there and never reads it)
    fr_a, fr_b = fr_a.copy(), fr_b.copy()
    for f in (fr_a, fr_b):
        f["msg"][f["nbits"] == 56, 7:] = 0
    for name in ("offset", "msg", "nbits", "errorbit", "pass", "phase_applied", "df", "addr"):
        if not np.array_equal(fr_a[name], fr_b[name]):
            i = int(np.nonzero(np.any(np.atleast_2d(fr_a[name] != fr_b[name]).reshape(len(fr_a), -1), axis=1))[0][0])
            raise AssertionError("frame field %s differs at %d: %r vs %r" % (name, i, fr_a[i], fr_b[i]))
    for name in ac_a.dtype.names:
        if not np.array_equal(ac_a[name], ac_b[name]):
            i = int(np.nonzero(ac_a[name] != ac_b[name])[0][0])
            raise AssertionError("aircraft field %s differs at %d: %r vs %r" % (name, i, ac_a[i], ac_b[i]))


def callback_text(ac):
    """The reference test's text stream (tests/test_1090.cpp:35-42) from aircraft snapshots."""
    return [O.format_aircraft(int(a["addr"]), a["callsign"].ljust(8, b"\0") if isinstance(a["callsign"], bytes) else a["callsign"],
                              int(a["lat1e7"]), int(a["lon1e7"]), int(a["altitude"]), int(a["speed"]), int(a["squawk"])) for a in ac]
