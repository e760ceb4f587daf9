"""UAT 978, CPU side: the oracle restatement against itself (parity unpinned, see oracle/oracle978.c), the product's
Reed-Solomon decoder (host code) against the oracle's, the generator round trip, and the re-buffering quirk of
UAT978.cpp:57."""
import numpy as np
import pytest

import libadsb_amd as A
from libadsb_amd import synth
from oracle import oracle_py as O

import uat_helpers as U

CODES = {0: (18, 12), 1: (34, 14), 2: (72, 20)}  # kind -> (data bytes, parity bytes)


def test_phase_lut_known_answers():
    lut = O.phase_lut978()
    # UAT978.cpp:86-97: index I | Q << 8, angle of (I - 127.5, Q - 127.5) + pi, scaled so that 2 pi = 65536, clamped
    assert lut[255 | (128 << 8)] == round(32768 * (np.arctan2(0.5, 127.5) + np.pi) / np.pi)
    assert lut[0 | (127 << 8)] == 41            # just below the negative real axis: angle close to 0 (+)
    assert lut[0 | (128 << 8)] == 65495         # just above it: close to 2 pi
    assert lut[128 | (255 << 8)] == round(32768 * (np.arctan2(127.5, 0.5) + np.pi) / np.pi)
    assert int(lut.max()) <= 65535 and int(lut.min()) >= 0
    # antisymmetry of atan2 about the centre 127.5: opposite points differ by half a turn
    i, q = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    a = lut[i | (q << 8)].astype(np.int64)
    b = lut[(255 - i) | ((255 - q) << 8)].astype(np.int64)
    assert np.all(np.abs(((a - b) % 65536) - 32768) <= 1)


def test_phase_lut_has_exactly_the_symmetries_the_demodulation_kernel_folds_it_by():
    """uat978.hip keeps the quadrant I, Q >= 128 in LDS (lut2_folded) and derives the rest: lut(255 - I, Q) = 32768 - lut(I, Q) and
    lut(I, 255 - Q) = -lut(I, Q), mod 65536, for every one of the 65 536 entries (the handle checks the same of its own table when it
    is made); no entry is near a rounding tie."""
    lut = O.phase_lut978().astype(np.int64)
    i, q = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    v = lut[i | (q << 8)]
    assert np.all((v + lut[i | ((255 - q) << 8)]) % 65536 == 0)
    assert np.all((v + lut[(255 - i) | (q << 8)]) % 65536 == 32768)
    x = 32768 * (np.arctan2(q - 127.5, i - 127.5) + np.pi) / np.pi
    assert np.min(np.abs(x - np.floor(x) - 0.5)) > 1e-4
    # the folded look-up itself, as the kernel does it: magnitudes, the quadrant's entry, the two corrections
    mi, mq = np.where(i >= 128, i - 128, 127 - i), np.where(q >= 128, q - 128, 127 - q)
    f = lut[(128 + mi) | ((128 + mq) << 8)]
    f = np.where((i >= 128) != (q >= 128), -f, f)
    f = np.where(i < 128, f + 32768, f) % 65536
    assert np.array_equal(f, v)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_rs_encode_decode_round_trip(kind):
    rng = np.random.default_rng(978 + kind)
    k, nr = CODES[kind]
    for trial in range(200):
        data = rng.integers(0, 256, k, dtype=np.uint8).tobytes()
        cw = bytearray(data + O.rs_parity978(kind, data))
        nerr = trial % (nr // 2 + 1)
        pos = rng.choice(k + nr, nerr, replace=False)
        for p in pos:
            cw[p] ^= int(rng.integers(1, 256))
        n, fixed = O.rs_decode978(kind, cw)
        assert n == nerr and fixed[:k] == data


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_product_rs_decoder_matches_oracle_including_beyond_capacity(native_libs, kind):
    """Same (count, bytes) for clean, correctable, uncorrectable and random words: outside the correction radius the
    result depends on the decoding procedure, so the two implementations must agree there too."""
    rng = np.random.default_rng(1187 + kind)
    k, nr = CODES[kind]
    for trial in range(1500):
        data = rng.integers(0, 256, k, dtype=np.uint8).tobytes()
        cw = bytearray(data + O.rs_parity978(kind, data))
        mode = trial % 5
        if mode == 4:
            cw = bytearray(rng.integers(0, 256, k + nr, dtype=np.uint8).tobytes())
        else:
            nerr = int(rng.integers(0, nr // 2 + 4)) if mode else trial % (nr // 2 + 1)
            for p in rng.choice(k + nr, min(nerr, k + nr), replace=False):
                cw[p] ^= int(rng.integers(1, 256))
        want = O.rs_decode978(kind, cw)
        got = A.rs_decode978(kind, cw)
        assert got == want, (trial, mode)


def test_generator_frames_are_recovered_by_the_oracle():
    cfg = synth.default_cfg978()
    iq, man = synth.fill978(3, 6 * 262144, cfg, manifest=True)
    out = O.Oracle978(carry_full=True).handle_data(iq)
    injected = {}
    for m in man:
        injected[int(m["start"])] = (bytes(m["data"][:m["len"]]), int(m["bad_bytes"]), int(m["kind"]))
    assert len(out) >= 0.9 * len(man) > 50
    for updown, payload, rs, idx in out:
        near = [s for s in (idx - 1, idx, idx + 1) if s in injected]
        assert near, "decoded frame at %d was not injected" % idx
        want, bad, kind = injected[near[0]]
        assert payload == want and updown == ("+" if kind == 2 else "-")
        assert rs <= bad  # a corrupted byte may fall on a byte the generator happened to leave unchanged


def test_reference_half_tail_carry_loses_frames_and_repeats_others():
    """UAT978.cpp:57 hands memmove the tail length in entries as a byte count, so the second half of the carried region
    keeps what the staging buffer held there: phases from one staging round earlier.  The oracle restates that; this
    test pins what it means against a full carry: frames starting in that region are lost, and frames that lay there one
    round earlier can be reported a second time, at a later stream index."""
    iq, man = synth.fill978(0, 4 * 262144, synth.default_cfg978(), manifest=True)
    ref = O.Oracle978(carry_full=False)
    full = O.Oracle978(carry_full=True)
    a, b = ref.handle_data(iq), full.handle_data(iq)
    assert ref.stream_state() == full.stream_state()
    assert {f[1] for f in a} <= {f[1] for f in b}          # nothing is invented ...
    assert len(set(b) - set(a)) > 0                        # ... some real frames are lost ...
    ghosts = sorted(set(a) - set(b), key=lambda f: f[3])   # ... and these are repeats of earlier frames
    real_index = {f[1]: f[3] for f in b}
    for updown, payload, rs, idx in ghosts:
        assert idx > real_index[payload] and (idx - real_index[payload]) % 2 == 0


def test_with_a_full_carry_the_cut_of_the_input_does_not_matter():
    """process_buffer keeps no state and asks for the tail again, so with the whole tail carried the frames do not depend
    on how HandleData's input is cut (in reference mode they do: the frozen half of the carried region differs)."""
    iq = synth.fill978(1, 8 * 262144, synth.default_cfg978())
    one = O.Oracle978(carry_full=True).handle_data(iq)
    assert len(one) > 100
    for cut in (262144, 100000, 65536 * 2 + 2):
        o = O.Oracle978(carry_full=True)
        many = []
        for k in range(0, iq.size, cut):
            many += o.handle_data(iq[k:k + cut])
        assert many == one, cut
    # reference mode on production-sized calls: every payload it reports is a real one
    ref = O.Oracle978().handle_data(iq)
    assert {f[1] for f in ref} <= {f[1] for f in one}


def stale_register_case(seed=5):
    """Phases of: idle, short frame, one more bit, then a long frame whose sync word lacks its first three bits."""
    rng = np.random.default_rng(seed)
    p1 = bytes([0x00]) + rng.integers(0, 256, 17, dtype=np.uint8).tobytes()
    p2 = bytes([0x08]) + rng.integers(0, 256, 33, dtype=np.uint8).tobytes()
    f1 = U.short_frame(p1)
    assert (f1[-1] & 3) != 3, "pick another seed: the overlap must differ from the sync prefix"
    bits = U.quiet_bits(rng, 64) + U.bits_of(U.ADSB_SYNC, 36) + U.bytes_to_bits(f1) + [0]
    bits += U.bits_of(U.ADSB_SYNC, 36)[3:] + U.bytes_to_bits(U.long_frame(p2)) + U.quiet_bits(rng, 2 * 4500)
    return U.phases_from_bits(bits), p1, p2


def test_stale_shift_registers_fire_on_a_sync_word_overlapping_the_previous_frame():
    """process_buffer jumps over a decoded frame (the loop resumes one bit after it) without clearing its two 18-bit
    registers.  The 18-bit check word starts and ends with 111, so a sync word whose first three bits are missing from
    the stream is still "found" through the stale bits, 15 bits after the jump; the 36-bit re-check tolerates four wrong
    bits and the frame decodes.  The stream itself holds no 18-bit match there."""
    phi, p1, p2 = stale_register_case()
    out, done = O.process_buffer978(phi)
    assert [(f[0], f[1]) for f in out] == [("-", p1), ("-", p2)]
    assert out[1][3] == out[0][3] + 2 * (36 + 240 + 1 - 3)
    assert done > 0
    # the same stream presented from just after the first frame: no stale bits, the second frame is not found
    cut = out[0][3] + 2 * (36 + 240) - 40
    out2, _ = O.process_buffer978(phi[cut:])
    assert out2 == []
