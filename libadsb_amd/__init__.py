"""libadsb_amd -- MI355X-native 1090ES IQ -> Mode S demodulator behind libadsb's handler surface.

This package is a thin ctypes view of libadsb_amd.so (include/adsb_amd.h).  The demodulation runs in
hand-written gfx950 HIP kernels; there is no Python or CPU fallback: if the shared library is missing,
or no HIP device is usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADSB_AMD_LIB") or os.path.join(_HERE, "libadsb_amd.so")  # override: A/B of build variants only

REF_BUFFER_BYTES = 262144
MODE_2000, MODE_2400 = 20, 24  # samples per microsecond x 10: the reference's demodulator / the library's own 2.4 MS/s mode
F_PASS2, F_PHASE, F_NEEDS_ICAO = 1, 2, 4

RECORD_DTYPE = np.dtype([("buffer", "<u4"), ("offset", "<u4"), ("addr", "<u4"), ("reserved", "<u2"), ("nbits", "u1"),
                         ("errorbit", "i1"), ("df", "u1"), ("flags", "u1"), ("msg", "u1", (14,))])
FRAME_DTYPE = np.dtype([("offset", "<u8"), ("msg", "u1", (14,)), ("nbits", "u1"), ("errorbit", "i1"), ("pass", "u1"),
                        ("phase_applied", "u1"), ("df", "u1"), ("reserved", "u1"), ("addr", "<u4")])
AIRCRAFT_DTYPE = np.dtype([("addr", "<u4"), ("callsign", "S8"), ("lat1e7", "<i4"), ("lon1e7", "<i4"), ("altitude", "<i4"),
                           ("speed", "<u4"), ("track", "<u4"), ("vert_rate", "<i4"), ("squawk", "<u4")])
DECODED_DTYPE = np.dtype([("kind", "u1"), ("metype", "u1"), ("mesub", "u1"), ("odd", "u1"), ("altitude", "<i4"), ("a", "<u4"), ("b", "<u4")])
PACKED_DTYPE = np.dtype([("buffer", "<u4"), ("offset", "<u4"), ("addr", "<u4"), ("reserved", "<u2"), ("nbits", "u1"), ("errorbit", "i1"),
                         ("df", "u1"), ("flags", "u1"), ("kind", "u1"), ("odd", "u1"), ("altitude", "<i4"), ("a", "<u4"), ("b", "<u4")])
OUT_RECORDS, OUT_DECODED, OUT_PACKED = 1, 2, 4
K_NONE, K_ALTITUDE, K_IDENT, K_POSITION, K_VELOCITY = range(5)
assert PACKED_DTYPE.itemsize == 32
assert RECORD_DTYPE.itemsize == 32 and FRAME_DTYPE.itemsize == 32 and AIRCRAFT_DTYPE.itemsize == 40 and DECODED_DTYPE.itemsize == 16

ON_CHANGED = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)

EXPORTS = [
    "adsb_amd_version", "adsb_amd_create", "adsb_amd_create_mode", "adsb_amd_handler_create_mode", "adsb_amd_resolver_set_mode", "adsb_amd_destroy", "adsb_amd_last_error", "adsb_amd_scan_1090",
    "adsb_amd_scan_1090_submit", "adsb_amd_scan_1090_fetch", "adsb_amd_scan_1090_fetch_decoded", "adsb_amd_scan_1090_fetch_packed", "adsb_amd_scan_1090_fetch_packed_begin", "adsb_amd_scan_1090_fetch_packed_end", "adsb_amd_set_outputs", "adsb_amd_resolver_feed_packed", "adsb_amd_handler_set_frames", "adsb_amd_scan_1090_fetch_device", "adsb_amd_scan_1090_fetch_device_packed", "adsb_amd_scan_1090_timing", "adsb_amd_set_timing", "adsb_amd_magnitude_1090",
    "adsb_amd_decode_1090", "adsb_amd_decode_record_host", "adsb_amd_resolver_feed_decoded", "adsb_amd_cpr_nl", "adsb_amd_cpr_global", "adsb_amd_cpr_global_batch",
    "adsb_amd_resolver_create", "adsb_amd_resolver_destroy", "adsb_amd_resolver_set_sample_clock", "adsb_amd_resolver_feed",
    "adsb_amd_resolver_aircraft_count", "adsb_amd_count_callback", "adsb_amd_handler_create", "adsb_amd_handler_destroy", "adsb_amd_handler_last_error",
    "adsb_amd_handler_set_sample_clock", "adsb_amd_handler_handle_data", "adsb_amd_handler_replay_file", "adsb_amd_handler_run_replay", "adsb_amd_host_alloc", "adsb_amd_host_free",
    "adsb_amd_shm_post_header", "adsb_amd_shm_read_header", "adsb_amd_shm_store_release", "adsb_amd_shm_load_acquire",
    "adsb_amd_transport_create", "adsb_amd_transport_destroy", "adsb_amd_transport_start", "adsb_amd_transport_stop", "adsb_amd_transport_push",
    "adsb_amd_transport_stats",
    "adsb_amd_uat_create", "adsb_amd_uat_destroy", "adsb_amd_uat_last_error", "adsb_amd_uat_handle_data", "adsb_amd_uat_set_carry_full", "adsb_amd_uat_set_host_loop", "adsb_amd_uat_set_extra_capacity",
    "adsb_amd_uat_stream_state", "adsb_amd_uat_process_phases", "adsb_amd_uat_process_iq", "adsb_amd_uat_submit_iq", "adsb_amd_uat_max_in_flight", "adsb_amd_uat_part_scan", "adsb_amd_uat_part_finish", "adsb_amd_uat_collect", "adsb_amd_uat_possible_steps", "adsb_amd_uat_check_word", "adsb_amd_uat_timing", "adsb_amd_uat_host_timing", "adsb_amd_uat_phase_lut",
    "adsb_amd_uat_rs_decode", "adsb_amd_uat_rs_decode_device", "adsb_amd_uat_set_dump_raw_message", "init_fec", "process_buffer",
]


class AdsbAmdError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libadsb_amd.so (never builds, never falls back)."""
    global _lib
    if _lib is None:
        # When PyTorch shares the process it must initialise its bundled HIP runtime first: both runtimes carry the
        # SONAME libamdhip64.so.7, and torch cannot find a device once the system runtime has been loaded before it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise AdsbAmdError("%s is missing: run `python -m libadsb_amd.build` (needs hipcc); there is no fallback path" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.adsb_amd_version.restype = C.c_char_p
        L.adsb_amd_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.adsb_amd_create_mode.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
        L.adsb_amd_handler_create_mode.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
        L.adsb_amd_resolver_set_mode.argtypes = [C.c_void_p, C.c_int]
        L.adsb_amd_destroy.argtypes = [C.c_void_p]
        L.adsb_amd_last_error.argtypes = [C.c_void_p]
        L.adsb_amd_last_error.restype = C.c_char_p
        L.adsb_amd_scan_1090.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.adsb_amd_scan_1090_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int]
        L.adsb_amd_scan_1090_fetch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.adsb_amd_scan_1090_fetch_decoded.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.adsb_amd_decode_1090.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        if hasattr(L, "adsb_amd_set_outputs"):  # absent from the older builds tools/ab.py compares against
            L.adsb_amd_set_outputs.argtypes = [C.c_void_p, C.c_uint]
            L.adsb_amd_scan_1090_fetch_packed.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
            if hasattr(L, "adsb_amd_scan_1090_fetch_packed_begin"):  # absent from the older builds tools/ab.py compares against (the export test covers the product)
                L.adsb_amd_scan_1090_fetch_packed_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
                L.adsb_amd_scan_1090_fetch_packed_end.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
            L.adsb_amd_resolver_feed_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
            L.adsb_amd_resolver_feed_packed.restype = C.c_long
            L.adsb_amd_handler_set_frames.argtypes = [C.c_void_p, C.c_int]
            L.adsb_amd_handler_set_frames.restype = None
            L.adsb_amd_scan_1090_fetch_device_packed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
        L.adsb_amd_decode_record_host.argtypes = [C.c_void_p, C.c_void_p]
        L.adsb_amd_decode_record_host.restype = None
        L.adsb_amd_cpr_nl.argtypes = [C.c_double]
        L.adsb_amd_cpr_global.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        if hasattr(L, "adsb_amd_cpr_global_batch"):  # absent from the older builds tools/ab.py compares against (the export test covers the product)
            L.adsb_amd_cpr_global_batch.argtypes = [C.c_size_t] + [C.c_void_p] * 8
            L.adsb_amd_cpr_global_batch.restype = None
        L.adsb_amd_resolver_feed_decoded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.adsb_amd_resolver_feed_decoded.restype = C.c_long
        L.adsb_amd_scan_1090_fetch_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
        L.adsb_amd_scan_1090_timing.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        if hasattr(L, "adsb_amd_set_timing"):  # absent from the older builds tools/ab.py compares against (the export test covers the product)
            L.adsb_amd_set_timing.argtypes = [C.c_void_p, C.c_uint]
        L.adsb_amd_magnitude_1090.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.adsb_amd_resolver_create.restype = C.c_void_p
        L.adsb_amd_resolver_destroy.argtypes = [C.c_void_p]
        L.adsb_amd_resolver_set_sample_clock.argtypes = [C.c_void_p, C.c_int64, C.c_uint32]
        L.adsb_amd_resolver_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.adsb_amd_resolver_feed.restype = C.c_long
        L.adsb_amd_resolver_aircraft_count.argtypes = [C.c_void_p]
        L.adsb_amd_resolver_aircraft_count.restype = C.c_size_t
        L.adsb_amd_handler_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.adsb_amd_handler_destroy.argtypes = [C.c_void_p]
        L.adsb_amd_handler_last_error.argtypes = [C.c_void_p]
        L.adsb_amd_handler_last_error.restype = C.c_char_p
        L.adsb_amd_handler_set_sample_clock.argtypes = [C.c_void_p, C.c_int64, C.c_uint32]
        L.adsb_amd_handler_handle_data.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.adsb_amd_handler_handle_data.restype = C.c_long
        L.adsb_amd_handler_replay_file.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.adsb_amd_handler_replay_file.restype = C.c_long
        L.adsb_amd_handler_run_replay.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.adsb_amd_handler_run_replay.restype = C.c_long
        L.adsb_amd_transport_create.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        L.adsb_amd_transport_destroy.argtypes = [C.c_void_p]
        L.adsb_amd_transport_start.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.adsb_amd_transport_stop.argtypes = [C.c_void_p]
        L.adsb_amd_transport_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.adsb_amd_transport_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.adsb_amd_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.adsb_amd_host_free.argtypes = [C.c_void_p]
        L.adsb_amd_shm_post_header.argtypes = [C.c_void_p] + [C.c_int64] * 4
        L.adsb_amd_shm_post_header.restype = None
        L.adsb_amd_shm_read_header.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.adsb_amd_shm_store_release.argtypes = [C.c_void_p, C.c_int64]
        L.adsb_amd_shm_store_release.restype = None
        L.adsb_amd_shm_load_acquire.argtypes = [C.c_void_p]
        L.adsb_amd_shm_load_acquire.restype = C.c_int64
        L.adsb_amd_uat_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.adsb_amd_uat_destroy.argtypes = [C.c_void_p]
        L.adsb_amd_uat_last_error.argtypes = [C.c_void_p]
        L.adsb_amd_uat_last_error.restype = C.c_char_p
        L.adsb_amd_uat_handle_data.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.adsb_amd_uat_set_carry_full.argtypes = [C.c_void_p, C.c_int]
        L.adsb_amd_uat_set_host_loop.argtypes = [C.c_void_p, C.c_int]
        L.adsb_amd_uat_set_extra_capacity.argtypes = [C.c_void_p, C.c_uint32]
        L.adsb_amd_uat_stream_state.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
        L.adsb_amd_uat_process_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.adsb_amd_uat_process_iq.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p,
                                              C.POINTER(C.c_int64)]
        L.adsb_amd_uat_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.adsb_amd_uat_phase_lut.argtypes = [C.c_void_p, C.c_void_p]
        L.adsb_amd_uat_host_timing.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 4
        L.adsb_amd_uat_submit_iq.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.adsb_amd_uat_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        if hasattr(L, "adsb_amd_uat_part_scan"):
            L.adsb_amd_uat_part_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
            L.adsb_amd_uat_part_finish.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p,
                                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.adsb_amd_uat_rs_decode.argtypes = [C.c_int, C.c_void_p]
        L.adsb_amd_uat_rs_decode_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.adsb_amd_uat_set_dump_raw_message.argtypes = [C.c_void_p]
        L.init_fec.restype = None
        L.process_buffer.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.process_buffer.restype = C.c_int
        _lib = L
    return _lib


class _Collector:
    """Turns the C callback stream into two structured arrays."""

    def __init__(self):
        self.frames, self.aircraft = [], []
        self.cb = ON_CHANGED(self._on)

    def _on(self, _user, frame, aircraft):
        self.frames.append(C.string_at(frame, 32))
        self.aircraft.append(C.string_at(aircraft, 40))

    def arrays(self):
        fr = np.frombuffer(b"".join(self.frames), dtype=FRAME_DTYPE) if self.frames else np.zeros(0, FRAME_DTYPE)
        ac = np.frombuffer(b"".join(self.aircraft), dtype=AIRCRAFT_DTYPE) if self.aircraft else np.zeros(0, AIRCRAFT_DTYPE)
        return fr, ac


class Scanner:
    """GPU half: u8 IQ -> sorted candidate records (adsb_amd_ctx_t)."""

    def __init__(self, device=-1, mode=MODE_2000):
        self._l = lib()
        h = C.c_void_p()
        rc = self._l.adsb_amd_create_mode(C.byref(h), device, mode)
        if rc != 0:
            raise AdsbAmdError("adsb_amd_create failed (%d): %s" % (rc, self._l.adsb_amd_last_error(None).decode()))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._l.adsb_amd_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise AdsbAmdError("libadsb_amd error %d: %s" % (rc, self._l.adsb_amd_last_error(self._h).decode()))

    def scan(self, iq, buffer_bytes=0):
        """Synchronous scan of a host uint8 array; returns a RECORD_DTYPE array."""
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        cap = max(1024, iq.size // 64)
        while True:
            out = np.empty(cap, dtype=RECORD_DTYPE)
            n = C.c_size_t(0)
            rc = self._l.adsb_amd_scan_1090(self._h, iq.ctypes.data, iq.size, buffer_bytes, out.ctypes.data, cap, C.byref(n))
            if rc == -4:
                cap = int(n.value)
                continue
            self._check(rc)
            return out[:n.value].copy()

    def submit(self, device_ptr, nbytes, buffer_bytes=REF_BUFFER_BYTES, stream=0, slot=0):
        self._check(self._l.adsb_amd_scan_1090_submit(self._h, C.c_void_p(device_ptr), nbytes, buffer_bytes, C.c_void_p(stream), slot))

    def fetch(self, slot=0, copy=True):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._l.adsb_amd_scan_1090_fetch(self._h, slot, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, RECORD_DTYPE)
        buf = (C.c_uint8 * (n.value * 32)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=RECORD_DTYPE)
        return arr.copy() if copy else arr

    def fetch_decoded(self, slot=0, copy=True):
        """fetch plus the GPU-decoded fields of every record: (records, decoded), parallel arrays."""
        p, q, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._check(self._l.adsb_amd_scan_1090_fetch_decoded(self._h, slot, C.byref(p), C.byref(q), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, RECORD_DTYPE), np.zeros(0, DECODED_DTYPE)
        rec = np.frombuffer((C.c_uint8 * (n.value * 32)).from_address(p.value), dtype=RECORD_DTYPE)
        dec = np.frombuffer((C.c_uint8 * (n.value * 16)).from_address(q.value), dtype=DECODED_DTYPE)
        return (rec.copy(), dec.copy()) if copy else (rec, dec)

    def set_outputs(self, mask):
        """Which arrays the ordering pass produces from the next submit on (OUT_RECORDS | OUT_DECODED | OUT_PACKED)."""
        self._check(self._l.adsb_amd_set_outputs(self._h, mask))

    def fetch_packed(self, slot=0, copy=True):
        """The packed hand-over form (record head + decoded fields, 32 bytes, no message bytes); needs set_outputs(OUT_PACKED)."""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._l.adsb_amd_scan_1090_fetch_packed(self._h, slot, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, PACKED_DTYPE)
        arr = np.frombuffer((C.c_uint8 * (n.value * 32)).from_address(p.value), dtype=PACKED_DTYPE)
        return arr.copy() if copy else arr

    def has_split_fetch(self):
        return hasattr(self._l, "adsb_amd_scan_1090_fetch_packed_begin")

    def fetch_packed_begin(self, slot=0):
        """First half of fetch_packed: waits for the slot's count, starts the copy and frees the slot for its next submit.  Returns the count."""
        n = C.c_size_t()
        self._check(self._l.adsb_amd_scan_1090_fetch_packed_begin(self._h, slot, C.byref(n)))
        return n.value

    def fetch_packed_end(self, slot=0, copy=True):
        """Second half: waits for the copy begun by fetch_packed_begin(slot)."""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._l.adsb_amd_scan_1090_fetch_packed_end(self._h, slot, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, PACKED_DTYPE)
        arr = np.frombuffer((C.c_uint8 * (n.value * 32)).from_address(p.value), dtype=PACKED_DTYPE)
        return arr.copy() if copy else arr

    def decode(self, records):
        """The ordering pass's field decoder run on the device over arbitrary records (parity helper)."""
        records = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
        out = np.zeros(len(records), dtype=DECODED_DTYPE)
        self._check(self._l.adsb_amd_decode_1090(self._h, records.ctypes.data, len(records), out.ctypes.data))
        return out

    def fetch_device(self, slot, dst_ptr, cap_records, stream=0, packed=False):
        """Waits for the slot and copies its records (packed: the packed form) to `dst_ptr` -- device memory or page-locked host
        memory -- enqueued on `stream`; returns the count."""
        n = C.c_size_t()
        fn = self._l.adsb_amd_scan_1090_fetch_device_packed if packed else self._l.adsb_amd_scan_1090_fetch_device
        self._check(fn(self._h, slot, C.c_void_p(dst_ptr), cap_records, C.c_void_p(stream), C.byref(n)))
        return n.value

    def timing(self, slot=0):
        a, b = C.c_float(), C.c_float()
        self._check(self._l.adsb_amd_scan_1090_timing(self._h, slot, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_timing(self, every):
        """HIP events around the demodulation kernel of every `every`-th submit (1: all, the default; 0: none)."""
        self._check(self._l.adsb_amd_set_timing(self._h, every))

    def timing_if_timed(self, slot=0):
        """(kernel ms, scan start -> count on the host ms) of the slot's last scan, or None when that scan carried no timing events."""
        a, b = C.c_float(), C.c_float()
        if self._l.adsb_amd_scan_1090_timing(self._h, slot, C.byref(a), C.byref(b)) != 0:
            return None
        return a.value, b.value

    def magnitude(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        out = np.empty(iq.size // 2, dtype=np.uint16)
        self._check(self._l.adsb_amd_magnitude_1090(self._h, iq.ctypes.data, iq.size, out.ctypes.data))
        return out


class Resolver:
    """Host half: records -> accepted frames + aircraft snapshots (adsb_amd_resolver_t).  Needs no GPU."""

    def __init__(self, sample_clock_hz=2000000, t0_ns=1_600_000_000 * 10**9, mode=MODE_2000):
        self._l = lib()
        self._h = C.c_void_p(self._l.adsb_amd_resolver_create())
        self._l.adsb_amd_resolver_set_sample_clock(self._h, t0_ns, sample_clock_hz)
        self._l.adsb_amd_resolver_set_mode(self._h, mode)

    def close(self):
        if getattr(self, "_h", None):
            self._l.adsb_amd_resolver_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def feed(self, records, samples_per_buffer, nbuffers, collect=True, count_callbacks=False, decoded=None):
        """collect: gather every callback's frame + aircraft snapshot (Python trampoline, slow); count_callbacks: fire the
        library's own counting listener instead (the callback path at native speed); neither: no callback at all.
        decoded: the GPU's decoded fields for these records (Scanner.fetch_decoded); None: the host decodes."""
        packed = getattr(records, "dtype", None) == PACKED_DTYPE  # the packed hand-over form: frames come back without message bytes
        records = np.ascontiguousarray(records, dtype=PACKED_DTYPE if packed else RECORD_DTYPE)
        col = _Collector()
        cb, user = (C.cast(col.cb, C.c_void_p), None) if collect else (None, None)
        counter = C.c_uint64(0)
        if count_callbacks and not collect:
            cb, user = C.cast(self._l.adsb_amd_count_callback, C.c_void_p), C.cast(C.pointer(counter), C.c_void_p)
        if packed:
            n = self._l.adsb_amd_resolver_feed_packed(self._h, records.ctypes.data, records.size, samples_per_buffer, nbuffers, cb, user)
        elif decoded is not None:
            decoded = np.ascontiguousarray(decoded, dtype=DECODED_DTYPE)
            assert len(decoded) == len(records)
            n = self._l.adsb_amd_resolver_feed_decoded(self._h, records.ctypes.data, decoded.ctypes.data, records.size, samples_per_buffer, nbuffers, cb, user)
        else:
            n = self._l.adsb_amd_resolver_feed(self._h, records.ctypes.data, records.size, samples_per_buffer, nbuffers, cb, user)
        if count_callbacks and not collect and n >= 0:
            assert counter.value == n, "one callback per accepted frame"
        if n < 0:
            raise AdsbAmdError("resolver_feed failed (%d)" % n)
        fr, ac = col.arrays()
        return n, fr, ac

    def aircraft_count(self):
        return self._l.adsb_amd_resolver_aircraft_count(self._h)


class Handler1090:
    """The reference's ADSB1090Handler surface (RTLSDR::IDataHandler::HandleData) over the GPU path."""

    def __init__(self, device=-1, sample_clock_hz=2000000, t0_ns=1_600_000_000 * 10**9, mode=MODE_2000):
        self._l = lib()
        h = C.c_void_p()
        rc = self._l.adsb_amd_handler_create_mode(C.byref(h), device, mode)
        if rc != 0:
            raise AdsbAmdError("adsb_amd_handler_create failed (%d): %s" % (rc, self._l.adsb_amd_last_error(None).decode()))
        self._h = h
        self._l.adsb_amd_handler_set_sample_clock(self._h, t0_ns, sample_clock_hz)

    def close(self):
        if getattr(self, "_h", None):
            self._l.adsb_amd_handler_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_frames(self, want_frames):
        """False: the listener looks at aircraft only; records travel in the packed form and frames carry no message bytes."""
        self._l.adsb_amd_handler_set_frames(self._h, 1 if want_frames else 0)

    def handle_data(self, iq, buffer_bytes=0, collect=True):
        """HandleData(span<const u8>): returns (frames, aircraft), one row per OnChanged callback, in callback order.
        collect=False: the library's counting listener takes the callbacks (native speed); returns the accepted-frame count."""
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        col = _Collector()
        counter = C.c_uint64(0)
        if collect:
            cb, user = C.cast(col.cb, C.c_void_p), None
        else:
            cb, user = C.cast(self._l.adsb_amd_count_callback, C.c_void_p), C.cast(C.pointer(counter), C.c_void_p)
        n = self._l.adsb_amd_handler_handle_data(self._h, iq.ctypes.data, iq.size, buffer_bytes, cb, user)
        if n < 0:
            raise AdsbAmdError("handle_data failed (%d): %s" % (n, self._l.adsb_amd_handler_last_error(self._h).decode()))
        if not collect:
            assert counter.value == n
            return n
        return col.arrays()

    def replay_file(self, path, first_buffer=0, max_buffers=2**62, collect=True):
        """RTLSDR::TestDataReadLoop for one pass over a recorded u8 IQ file (whole 262144-B buffers).  Returns
        (accepted frame count, frames, aircraft)."""
        col = _Collector()
        counter = C.c_uint64(0)
        if collect:
            cb, user = C.cast(col.cb, C.c_void_p), None
        else:  # the library's counting listener: the callback path at native speed
            cb, user = C.cast(self._l.adsb_amd_count_callback, C.c_void_p), C.cast(C.pointer(counter), C.c_void_p)
        n = self._l.adsb_amd_handler_replay_file(self._h, os.fsencode(path), first_buffer, max_buffers, cb, user)
        if n < 0:
            raise AdsbAmdError("replay_file failed (%d): %s" % (n, self._l.adsb_amd_handler_last_error(self._h).decode()))
        fr, ac = col.arrays()
        return n, fr, ac


    def run_replay(self, path, collect=True):
        """libadsb's replay mode in native code: the file through the 16-slot ring, one HandleData per 262144-byte slot on the
        transport's consumer thread.  Returns (accepted frames, frames, aircraft, slots delivered, seconds)."""
        col = _Collector()
        counter = C.c_uint64(0)
        if collect:
            cb, user = C.cast(col.cb, C.c_void_p), None
        else:
            cb, user = C.cast(self._l.adsb_amd_count_callback, C.c_void_p), C.cast(C.pointer(counter), C.c_void_p)
        nb, sec = C.c_uint64(), C.c_double()
        n = self._l.adsb_amd_handler_run_replay(self._h, os.fsencode(path), cb, user, C.byref(nb), C.byref(sec))
        if n < 0:
            raise AdsbAmdError("run_replay failed (%d): %s" % (n, self._l.adsb_amd_handler_last_error(self._h).decode()))
        fr, ac = col.arrays()
        return n, fr, ac, nb.value, sec.value


BUFFER_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)


class Transport:
    """The stand-alone sample transport (transport.hpp): 16 x 262144-byte ring, producer = file replay or push(), consumer thread
    -> sink(bytes).  Needs no GPU (the slots are page-locked only when a HIP runtime is usable)."""

    def __init__(self, replay_path=None, loop=False):
        self._l = lib()
        self._h = C.c_void_p()
        rc = self._l.adsb_amd_transport_create(C.byref(self._h), os.fsencode(replay_path) if replay_path else None, 1 if loop else 0)
        if rc != 0:
            raise AdsbAmdError("adsb_amd_transport_create failed (%d)" % rc)
        self._cb = None

    def start(self, sink):
        self._cb = BUFFER_FN(lambda _u, p, n: sink(C.string_at(p, n)))
        rc = self._l.adsb_amd_transport_start(self._h, C.cast(self._cb, C.c_void_p), None)
        if rc != 0:
            raise AdsbAmdError("adsb_amd_transport_start failed (%d): cannot open the recording" % rc)

    def push(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        rc = self._l.adsb_amd_transport_push(self._h, data.ctypes.data, data.size)
        if rc != 0:
            raise AdsbAmdError("adsb_amd_transport_push failed (%d): size must be a multiple of 262144" % rc)

    def stats(self):
        d, done, pl = C.c_uint64(), C.c_int(), C.c_int()
        self._l.adsb_amd_transport_stats(self._h, C.byref(d), C.byref(done), C.byref(pl))
        return {"delivered": d.value, "producer_done": bool(done.value), "page_locked": bool(pl.value)}

    def stop(self):
        self._l.adsb_amd_transport_stop(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._l.adsb_amd_transport_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class PinnedBuffer:
    """Page-locked host bytes as a numpy array (adsb_amd_host_alloc): a ring slot HandleData can upload from by DMA."""

    def __init__(self, nbytes):
        self._l = lib()
        self._p = C.c_void_p()
        if self._l.adsb_amd_host_alloc(C.byref(self._p), nbytes) != 0:
            raise AdsbAmdError("adsb_amd_host_alloc(%d) failed" % nbytes)
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self._p.value))

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            self._l.adsb_amd_host_free(self._p)
            self._p = None

    def __del__(self):
        self.close()


UAT_FRAME = C.CFUNCTYPE(None, C.c_void_p, C.c_char, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_uint64)
DUMP_RAW_MESSAGE = C.CFUNCTYPE(None, C.c_char, C.POINTER(C.c_uint8), C.c_int, C.c_int)


def pack_records(records, decoded):
    """The packed hand-over form from records and their decoded fields (what the ordering pass writes with OUT_PACKED)."""
    out = np.zeros(len(records), dtype=PACKED_DTYPE)
    for k in ("buffer", "offset", "addr", "reserved", "nbits", "errorbit", "df", "flags"):
        out[k] = records[k]
    for k in ("kind", "odd", "altitude", "a", "b"):
        out[k] = decoded[k]
    return out


def decode_records_host(records):
    """Host build of the field decoder (decode1090.h), record by record.  Needs no GPU."""
    records = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
    out = np.zeros(len(records), dtype=DECODED_DTYPE)
    L = lib()
    for i in range(len(records)):
        L.adsb_amd_decode_record_host(records.ctypes.data + 32 * i, out.ctypes.data + 16 * i)
    return out


def rs_decode978(kind, codeword):
    """The product's Reed-Solomon decoder (host code, no GPU): kind 0 RS(30,18), 1 RS(48,34), 2 RS(92,72).  Returns (count, bytes)."""
    buf = np.array(np.frombuffer(bytes(codeword), dtype=np.uint8))
    n = lib().adsb_amd_uat_rs_decode(kind, buf.ctypes.data)
    return n, buf.tobytes()


class Uat978:
    """UAT 978 path (UAT978Handler / process_buffer).  Frames come back as (updown, payload bytes, rs_errors, sample_index)."""

    def __init__(self, device=-1, carry_full=False):
        self._l = lib()
        self._h = C.c_void_p()
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device < 0 else device
        rc = self._l.adsb_amd_uat_create(C.byref(self._h), dev)
        if rc != 0:
            self._h = None
            raise AdsbAmdError("adsb_amd_uat_create failed (%d): %s" % (rc, self._l.adsb_amd_uat_last_error(None).decode()))
        if carry_full:
            self._l.adsb_amd_uat_set_carry_full(self._h, 1)

    def close(self):
        if getattr(self, "_h", None):
            self._l.adsb_amd_uat_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise AdsbAmdError("libadsb_amd UAT error %d: %s" % (rc, self._l.adsb_amd_uat_last_error(self._h).decode()))

    @staticmethod
    def _collector(out):
        def _cb(_u, updown, data, n, rs, idx):
            out.append((updown.decode(), bytes(data[:n]), int(rs), int(idx)))
        return UAT_FRAME(_cb)

    def handle_data(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        out = []
        cb = self._collector(out)
        self._check(self._l.adsb_amd_uat_handle_data(self._h, iq.ctypes.data, iq.size, cb, None))
        return out

    def set_host_loop(self, on):
        """on: the host walks the dump978 scan loop over the device's records (rounds 1-2); off (default): decided on the device"""
        self._check(self._l.adsb_amd_uat_set_host_loop(self._h, 1 if on else 0))

    def set_extra_capacity(self, entries):
        """entries (<= 4096) of the device's side array for frames reached through stale register bits; a call that needs more falls back to the host loop"""
        self._check(self._l.adsb_amd_uat_set_extra_capacity(self._h, int(entries)))

    def stream_state(self):
        off, used = C.c_uint64(), C.c_size_t()
        self._l.adsb_amd_uat_stream_state(self._h, C.byref(off), C.byref(used))
        return off.value, used.value

    def process_phases(self, phi, offset=0):
        phi = np.ascontiguousarray(phi, dtype=np.uint16)
        out, done = [], C.c_int64()
        cb = self._collector(out)
        self._check(self._l.adsb_amd_uat_process_phases(self._h, phi.ctypes.data, phi.size, offset, cb, None, C.byref(done)))
        return out, done.value

    def process_iq(self, iq, offset=0):
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        out, done = [], C.c_int64()
        cb = self._collector(out)
        self._check(self._l.adsb_amd_uat_process_iq(self._h, iq.ctypes.data, iq.size // 2, 0, offset, cb, None, C.byref(done)))
        return out, done.value

    def process_device(self, device_ptr, nsamples, offset=0, collect=True):
        out, done = [], C.c_int64()
        cb = self._collector(out) if collect else None
        self._check(self._l.adsb_amd_uat_process_iq(self._h, C.c_void_p(device_ptr), nsamples, 1, offset, cb, None, C.byref(done)))
        return out, done.value

    def submit_device(self, device_ptr, nsamples, offset=0):
        """GPU half of process_device on a worker thread (two calls may be in flight); collect() finishes the oldest."""
        self._check(self._l.adsb_amd_uat_submit_iq(self._h, C.c_void_p(device_ptr), nsamples, offset))

    def part_scan(self, device_ptr, nsamples):
        """The part of a cut stream's work that does not depend on the parts before it (see adsb_amd_uat_part_scan)."""
        self._check(self._l.adsb_amd_uat_part_scan(self._h, C.c_void_p(device_ptr), nsamples))

    def part_finish(self, own_begin, own_end, entry_bit, last, offset=0, collect=True):
        """-> (frames of the part's own start bits, exit bit, consumed); positions in samples / bits of the window given to part_scan."""
        out, exit_bit, done = [], C.c_int64(), C.c_int64()
        cb = self._collector(out) if collect else None
        self._check(self._l.adsb_amd_uat_part_finish(self._h, C.c_int64(own_begin), C.c_int64(own_end), C.c_int64(entry_bit), 1 if last else 0,
                                                     C.c_uint64(offset), cb, None, C.byref(exit_bit), C.byref(done)))
        return out, exit_bit.value, done.value

    def max_in_flight(self):
        return int(self._l.adsb_amd_uat_max_in_flight()) if hasattr(self._l, "adsb_amd_uat_max_in_flight") else 3

    def collect(self, collect=True):
        out, done = [], C.c_int64()
        cb = self._collector(out) if collect else None
        self._check(self._l.adsb_amd_uat_collect(self._h, cb, None, C.byref(done)))
        return out, done.value

    def timing(self):
        a, b, c, d = C.c_float(), C.c_float(), C.c_uint64(), C.c_uint64()
        self._l.adsb_amd_uat_timing(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        w = [C.c_float() for _ in range(4)]
        self._l.adsb_amd_uat_host_timing(self._h, *[C.byref(x) for x in w])
        return {"scan_ms": a.value, "demod_ms": b.value, "candidates": c.value, "extra_lookups": d.value,
                "host_wall_ms": {"match": w[0].value, "demod": w[1].value, "decide_kernels": w[2].value, "loop": w[3].value}}

    def rs_decode_device(self, kind, words):
        """words: (count, 30 | 48 | 92) uint8 -> (results int32[count], corrected words)."""
        w = np.ascontiguousarray(words, dtype=np.uint8).copy()
        res = np.empty(w.shape[0], dtype=np.int32)
        self._check(self._l.adsb_amd_uat_rs_decode_device(self._h, kind, w.ctypes.data, w.shape[0], res.ctypes.data))
        return res, w

    def phase_lut(self):
        lut = np.empty(65536, dtype=np.uint16)
        self._check(self._l.adsb_amd_uat_phase_lut(self._h, lut.ctypes.data))
        return lut
