"""The 2.4 MS/s scan mode, CPU side: its specification (oracle/oracle2400.c) recovers what the generator transmits.

No reference exists for this mode: parity unpinned.  What can be shown without one is self-consistency:
encode -> sample at 2.4 MS/s -> decode gives the transmitted bytes back, at the sample the frame starts in, and noise decodes to nothing."""
import numpy as np
import pytest

import libadsb_amd as A
from libadsb_amd import synth
from oracle import oracle_py as O

BB = A.REF_BUFFER_BYTES


def round_trip(records, frames, nsamples):
    """(transmitted frames far enough from the buffer end, how many of them were decoded at the right sample, unconditional records that are no transmitted message)"""
    got = {}
    for r in records:
        got.setdefault(bytes(r["msg"][:r["nbits"] // 8]), []).append(int(r["offset"]))
    truth, total, hit = set(), 0, 0
    for f in frames:
        nb = int(f["nbits"]) // 8
        msg = bytes(f["msg"][:nb])
        cand = [msg]
        fb = int(f["flipped_bit"])
        if fb >= 0:  # transmitted with one bit flipped: a DF11/17 comes back repaired
            m = bytearray(msg)
            m[fb >> 3] ^= 0x80 >> (fb & 7)
            cand.append(bytes(m))
        truth.update(cand)
        if int(f["start"]) + 300 >= nsamples:
            continue
        total += 1
        hit += any(c in got and any(abs(o - int(f["start"])) <= 1 for o in got[c]) for c in cand)
    false17 = sum(1 for r in records if (r["flags"] & A.F_NEEDS_ICAO) == 0 and bytes(r["msg"][:r["nbits"] // 8]) not in truth)
    return total, hit, false17


@pytest.mark.parametrize("over,floor", [(dict(), 0.97), (dict(noise_amp=1, amp_lo=100, amp_hi=128), 0.98), (dict(noise_amp=10), 0.80),
                                        (dict(pct_bitflip=100, pct_df17=60, pct_df11=40), 0.90)])
def test_specification_recovers_the_transmitted_frames(over, floor):
    cfg = synth.default_cfg(**over)
    total = hit = false17 = 0
    for b in range(8):
        iq, fr = synth.fill(b, BB, cfg, manifest=True, rate_x10=24)
        rec = O.expected_records2400(iq, BB, dtype=A.RECORD_DTYPE)
        assert np.all(np.diff(rec["offset"].astype(np.int64)) > 0) and np.all(rec["reserved"] < 5)
        t, h, f = round_trip(rec, fr, BB // 2)
        total, hit, false17 = total + t, hit + h, false17 + f
    assert total > 300 and hit >= floor * total, (hit, total)
    assert false17 == 0


def test_noise_decodes_to_nothing_and_rate_20_is_untouched():
    for amp in (3, 25, 60):
        iq, _ = synth.fill_range(0, 6, cfg=synth.default_cfg(mean_spacing=0, noise_amp=amp), rate_x10=24)
        rec = O.expected_records2400(iq, BB, dtype=A.RECORD_DTYPE)
        assert np.count_nonzero((rec["flags"] & A.F_NEEDS_ICAO) == 0) == 0
    a, na = synth.fill_range(3, 4)
    b, nb = synth.fill_range(3, 4, rate_x10=20)
    assert na == nb and np.array_equal(a, b)
    assert len(O.expected_records2400(a[:500], 0, dtype=A.RECORD_DTYPE)) == 0  # shorter than one window
