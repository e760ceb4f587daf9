"""The stand-alone slice of libadsb's surface (include/libadsb_iface.hpp) against the reference's own headers, at compile time.

Build container only: /root/reference never travels to the GPU box, so the test skips where it is absent (and it needs no GPU).  The
translation unit tests/cpp/iface_matches_reference.cpp includes /root/reference/ADSBListener.h and AircraftImpl.h (they need only
CommonMacros.h: no <rtl-sdr.h>, no patch) inside `namespace ref` beside the stand-alone header and static_asserts every shared name:
Source's enumerators, every IAirCraft / IListener / IDataProvider member-function signature, offsetof / sizeof of AirCraftImpl's
members, TrafficManager::FindOrCreate / SetListener / NotifyChanged and its two data members; the program it builds then calls every
virtual function of the stand-alone types through the reference's types, which pins the order of the virtual functions.
Reference: ADSBListener.h:27-72, AircraftImpl.h:9-68 (ADSB.h:10-28 pulls RTLSDR.hpp -> <rtl-sdr.h> and cannot be compiled in this image).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ADSBListener.h")), reason="the reference tree is not on this machine")
def test_standalone_interface_equals_the_reference_headers(tmp_path):
    exe = str(tmp_path / "iface_matches_reference")
    cmd = ["g++", "-std=c++20", "-Wall", "-Wextra", "-Wno-invalid-offsetof", "-I" + os.path.join(ROOT, "include"), "-I" + REF,
           os.path.join(ROOT, "tests", "cpp", "iface_matches_reference.cpp"), "-o", exe]
    cc = subprocess.run(cmd, capture_output=True, text=True)
    assert cc.returncode == 0, "libadsb_iface.hpp no longer matches the reference headers:\n" + cc.stderr[-4000:]
    run = subprocess.run([exe], capture_output=True, text=True)
    assert run.returncode == 0 and "iface ok" in run.stdout, run.stdout + run.stderr
