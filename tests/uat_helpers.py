"""Builders for hand-made UAT phase streams (tests only)."""
import numpy as np

from oracle import oracle_py as O

ADSB_SYNC = 0xEACDDA4E2
UPLINK_SYNC = 0x153225B1D
STEP = 6000  # phase step per sample, in LUT units (65536 = one turn)


def bits_of(value, nbits):
    return [(value >> (nbits - 1 - k)) & 1 for k in range(nbits)]


def bytes_to_bits(data):
    return [(b >> (7 - k)) & 1 for b in data for k in range(8)]


def short_frame(payload18):
    assert len(payload18) == 18 and (payload18[0] >> 3) == 0
    return bytes(payload18) + O.rs_parity978(0, payload18)


def long_frame(payload34):
    assert len(payload34) == 34 and (payload34[0] >> 3) != 0
    return bytes(payload34) + O.rs_parity978(1, payload34)


def uplink_frame(payload432):
    assert len(payload432) == 432
    out = bytearray(552)
    for block in range(6):
        d = payload432[block * 72:(block + 1) * 72]
        cw = bytes(d) + O.rs_parity978(2, d)
        for i in range(92):
            out[i * 6 + block] = cw[i]
    return bytes(out)


def phases_from_bits(bits, first=1000, samples_before=0, step=STEP):
    """Two samples per bit, the phase moves +-step per sample.  `samples_before` extra samples shift the alignment."""
    d = np.repeat(np.where(np.asarray(bits, dtype=np.int64) > 0, step, -step), 2)
    if samples_before:
        d = np.concatenate([np.full(samples_before, -step, dtype=np.int64), d])
    phi = (first + np.concatenate([[0], np.cumsum(d)])) & 0xFFFF
    return phi.astype(np.uint16)


def quiet_bits(rng, n):
    """Idle filler that cannot contain an 18-bit sync prefix: period-2 pattern with rare flips would; use 0011 runs."""
    base = np.tile([0, 0, 1, 1], n // 4 + 1)[:n]
    return list(base)
