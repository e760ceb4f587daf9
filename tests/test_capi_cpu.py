"""CPU-side checks of the C ABI: symbols, loud failure without a device, host resolver against the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import libadsb_amd as A
from libadsb_amd import synth
from oracle import oracle_py as O

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BB = A.REF_BUFFER_BYTES


def test_library_exports_every_declared_symbol(native_libs):
    hdr = open(os.path.join(ROOT, "include", "adsb_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(adsb_amd_[a-z0-9_]+)\s*\(", hdr)) - {"adsb_amd_on_changed_fn"}
    declared = sorted(declared | set(re.findall(r"\b(init_fec|process_buffer)\s*\(", hdr)))  # the reference's own UAT seam names
    assert declared == sorted(A.EXPORTS), "python binding and header disagree"
    lib = C.CDLL(native_libs[0])
    for name in declared:
        assert hasattr(lib, name), name


def test_record_layout_matches_header():
    assert A.RECORD_DTYPE.itemsize == 32
    assert [A.RECORD_DTYPE.fields[n][1] for n in ("buffer", "offset", "addr", "reserved", "nbits", "errorbit", "df", "flags", "msg")] == \
        [0, 4, 8, 12, 14, 15, 16, 17, 18]


def test_create_fails_loudly_without_a_device(native_libs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(A.AdsbAmdError, match="no usable HIP device"):
        A.Scanner()
    with pytest.raises(A.AdsbAmdError, match="adsb_amd_handler_create failed"):
        A.Handler1090()
    with pytest.raises(A.AdsbAmdError, match="adsb_amd_uat_create failed"):
        A.Uat978()
    # argument checks that need no device
    L = A.lib()
    assert L.adsb_amd_uat_create(None, 0) == -2 and L.adsb_amd_uat_handle_data(None, None, 0, None, None) == -2
    assert L.adsb_amd_create(None, 0) == -2 and L.adsb_amd_handler_replay_file(None, b"x", 0, 1, None, None) == -2
    with pytest.raises(A.AdsbAmdError):
        A.Handler1090()


@pytest.mark.parametrize("over", [dict(), dict(noise_amp=20), dict(mean_spacing=300, pool_size=16), dict(pct_df17=20, pct_df11=10)])
def test_resolver_applies_reference_sequencing(native_libs, over):
    # records as the GPU contract defines them (derived from the oracle's state-free probes) -> resolver
    # must reproduce the reference's sequential loop: skip-ahead, ICAO gating, retry order, CPR, callbacks
    iq, _ = synth.fill_range(20, 6, cfg=synth.default_cfg(**over))
    rec = H.expected_records(iq, BB)
    r = A.Resolver()
    n, fr, ac = r.feed(rec, BB // 2, 6)
    ofr, oac = H.oracle_run(iq, BB)
    assert n == len(ofr) > 50
    H.assert_streams_equal(fr, ac, ofr, oac)
    assert H.callback_text(ac) == H.callback_text(oac)


def test_resolver_state_carries_across_calls_and_ttl_expires(native_libs):
    # ICAO cache entries expire after 60 s of stream time (ADSB1090.cpp:200-207): feed AP-type-heavy buffers with a
    # gap in the sample clock and compare with the oracle driven identically
    cfg = synth.default_cfg(pct_df17=10, pct_df11=10, pool_size=8, mean_spacing=600)
    iq, _ = synth.fill_range(0, 4, cfg=cfg)
    r, o = A.Resolver(), O.Oracle1090()
    gap = np.full(BB, 127, dtype=np.uint8)
    total = 0
    for step in range(4):
        chunk = iq[step * BB:(step + 1) * BB]
        n, fr, ac = r.feed(H.expected_records(chunk, BB), BB // 2, 1)
        ofr, oac = o.handle_data(chunk)
        H.assert_streams_equal(fr, ac, ofr, oac)
        total += n
        if step == 1:  # 70 s of silence: 70 s * 2 MS/s = 1068 buffers of 131072 samples
            r.feed(np.zeros(0, A.RECORD_DTYPE), BB // 2, 1069)
            for _ in range(1069):
                o.handle_data(gap, collect=False)
    assert total > 100


def test_decode_known_frames_through_resolver(native_libs):
    # outputs the survey recorded from the reference itself
    import json
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_appendix_b.json")))
    r = A.Resolver()
    off = 1000
    last = None
    for v in vec["frames"]:
        msg = bytes.fromhex(v["hex"])
        rec = np.zeros(1, A.RECORD_DTYPE)
        rec["offset"] = off
        rec["msg"][0, :len(msg)] = np.frombuffer(msg, np.uint8)
        rec["nbits"] = len(msg) * 8
        rec["errorbit"] = -1
        rec["df"] = msg[0] >> 3
        rec["addr"] = int.from_bytes(msg[1:4], "big")
        n, fr, ac = r.feed(rec, BB // 2, 1)
        assert n == 1
        last = ac[0]
        for k, want in v["expect"].items():
            got = last[k]
            if k == "callsign":
                got = got.decode()
                got = got.ljust(8)
            assert got == want, (v["hex"], k, got, want)


def test_cxx_factories_are_exported(native_libs):
    # the two factories libadsb's callers bind (reference ADSB.h:13-15, 24-26), same namespaces and signatures
    import subprocess
    syms = subprocess.run(["nm", "-DC", native_libs[0]], capture_output=True, text=True).stdout
    for name in ("ADSB::TryCreateUAT978Handler", "ADSB::test::TryCreateUAT978Handler", "ADSB::GetThreadLocalTrafficManager"):
        assert name in syms, name
    assert "ADSB::TryCreateADSB1090Handler(std::shared_ptr<ADSB::TrafficManager> const&, RTLSDR::IDeviceSelector const*, ADSB::Source)" in syms
    assert "ADSB::test::TryCreateADSB1090Handler(std::shared_ptr<ADSB::TrafficManager> const&, RTLSDR::IDeviceSelector const*, ADSB::Source)" in syms
