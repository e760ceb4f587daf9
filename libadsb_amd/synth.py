"""Deterministic synthetic 1090ES u8 IQ -- ctypes view of libadsb_synth.so."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libadsb_synth.so")
_SRC = os.path.join(_HERE, "csrc", "synth1090.c")

BUFFER_BYTES = 262144  # RTLSDR::BufferLength (reference RTLSDR.hpp:55)


class SynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("noise_amp", C.c_int32), ("mean_spacing", C.c_int32), ("amp_lo", C.c_int32),
                ("amp_hi", C.c_int32), ("pct_df17", C.c_int32), ("pct_df11", C.c_int32), ("pct_bitflip", C.c_int32),
                ("pct_halfsample", C.c_int32), ("pool_size", C.c_int32)]


class Synth978Cfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("noise_amp", C.c_int32), ("amp_lo", C.c_int32), ("amp_hi", C.c_int32),
                ("mean_gap_bits", C.c_int32), ("pct_uplink", C.c_int32), ("pct_long", C.c_int32), ("pct_corrupt", C.c_int32),
                ("max_bad_bytes", C.c_int32)]


FRAME978_DTYPE = np.dtype([("start", "<u8"), ("kind", "u1"), ("bad_bytes", "u1"), ("len", "<u2"), ("data", "u1", (432,)), ("pad", "u1", (4,))])

FRAME_DTYPE = np.dtype([("start", "<u4"), ("msg", "u1", (14,)), ("nbits", "u1"), ("flipped_bit", "i1"),
                        ("half_sample", "u1"), ("amplitude", "u1"), ("pad", "u1", (2,))])  # sizeof(adsb_synth_frame_t) = 24
assert FRAME_DTYPE.itemsize == 24


def build(force=False):
    from . import build as _b
    return _b.build_synth(force)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.adsb_synth_default.argtypes = [C.POINTER(SynthCfg)]
        L.adsb_synth_fill.argtypes = [C.POINTER(SynthCfg), C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.adsb_synth_fill_range.argtypes = [C.POINTER(SynthCfg), C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int]
        L.adsb_synth_fill_range.restype = C.c_long
        L.adsb_synth_fill_range_rate.argtypes = [C.POINTER(SynthCfg), C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        L.adsb_synth_fill_range_rate.restype = C.c_long
        L.adsb_synth_fill_rate.argtypes = [C.POINTER(SynthCfg), C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
        L.adsb_synth_pool_addr.argtypes = [C.POINTER(SynthCfg), C.c_uint32]
        L.adsb_synth_pool_addr.restype = C.c_uint32
        L.adsb_synth978_default.argtypes = [C.POINTER(Synth978Cfg)]
        L.adsb_synth978_fill.argtypes = [C.POINTER(Synth978Cfg), C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_long]
        L.adsb_synth978_fill.restype = C.c_long
        _lib = L
    return _lib


def default_cfg(**over):
    c = SynthCfg()
    lib().adsb_synth_default(C.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def fill(buf_index, nbytes=BUFFER_BYTES, cfg=None, manifest=False, rate_x10=20):
    cfg = cfg or default_cfg()
    out = np.empty(nbytes, dtype=np.uint8)
    if manifest:
        fr = np.zeros(4096, dtype=FRAME_DTYPE)
        n = lib().adsb_synth_fill_rate(C.byref(cfg), buf_index, out.ctypes.data, nbytes, fr.ctypes.data, fr.size, rate_x10)
        return out, fr[:min(n, fr.size)]
    lib().adsb_synth_fill_rate(C.byref(cfg), buf_index, out.ctypes.data, nbytes, None, 0, rate_x10)
    return out


def fill_range(first_buf, nbuf, buf_bytes=BUFFER_BYTES, cfg=None, nthreads=None, out=None, rate_x10=20):
    """Returns (uint8 array of nbuf*buf_bytes, frames injected).  rate_x10 = 24: the same pulse trains sampled at 2.4 MS/s."""
    cfg = cfg or default_cfg()
    if out is None:
        out = np.empty(nbuf * buf_bytes, dtype=np.uint8)
    nthreads = nthreads or min(32, os.cpu_count() or 1)
    if rate_x10 == 20:
        n = lib().adsb_synth_fill_range(C.byref(cfg), first_buf, nbuf, out.ctypes.data, buf_bytes, nthreads)
    else:
        n = lib().adsb_synth_fill_range_rate(C.byref(cfg), first_buf, nbuf, out.ctypes.data, buf_bytes, nthreads, rate_x10)
    return out, n


def default_cfg978(**over):
    c = Synth978Cfg()
    lib().adsb_synth978_default(C.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def fill978(stream_index, nbytes, cfg=None, manifest=False):
    """Synthetic UAT 978 u8 IQ (2 samples per bit).  Returns the array, and the frame manifest when asked."""
    cfg = cfg or default_cfg978()
    out = np.empty(nbytes, dtype=np.uint8)
    if manifest:
        fr = np.zeros(8192, dtype=FRAME978_DTYPE)
        n = lib().adsb_synth978_fill(C.byref(cfg), stream_index, out.ctypes.data, nbytes, fr.ctypes.data, fr.size)
        return out, fr[:min(n, fr.size)]
    lib().adsb_synth978_fill(C.byref(cfg), stream_index, out.ctypes.data, nbytes, None, 0)
    return out
