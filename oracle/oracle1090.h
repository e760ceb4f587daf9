/*
 * oracle1090.h -- CPU restatement of the reference's 1090ES IQ->message->aircraft path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under libadsb_amd/ (the product) may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * Pin status: see the header of oracle1090.c.
 */
#ifndef ORACLE1090_H
#define ORACLE1090_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle1090 oracle1090_t;

/* One accepted Mode S frame, as the reference has it at ADSB1090.cpp:937 (UseModesMessage). */
typedef struct oracle1090_frame
{
    uint64_t offset;        /* sample index j of the preamble inside the HandleData buffer */
    uint8_t  msg[14];       /* Message::msg after CRC repair */
    uint8_t  nbits;         /* Message::msgbits (56 / 112) */
    int8_t   errorbit;      /* Message::errorbit */
    uint8_t  pass;          /* 1 = first slice, 2 = retry (useCorrection) */
    uint8_t  phase_applied; /* retry actually rescaled the window (DetectOutOfPhase != 0 and j != 0) */
    uint8_t  df;            /* Message::msgtype */
    uint8_t  reserved;
    uint32_t addr;          /* (aa1<<16)|(aa2<<8)|aa3 */
} oracle1090_frame_t;

/* Aircraft state snapshot as seen by IListener::OnChanged (AircraftImpl.h:9-47). */
typedef struct oracle1090_aircraft
{
    uint32_t addr;
    char     callsign[8];
    int32_t  lat1e7, lon1e7;
    int32_t  altitude;
    uint32_t speed, track;
    int32_t  vert_rate;
    uint32_t squawk; /* modeA: never written by the 1090 path, stays 0 */
} oracle1090_aircraft_t;

typedef void (*oracle1090_cb)(void* user, const oracle1090_frame_t* frame, const oracle1090_aircraft_t* aircraft);

/* Per-offset, state-free view of what the demodulator computes at sample j (both slicing passes).
 * Used to check the GPU's candidate records one by one. */
typedef struct oracle1090_probe
{
    uint8_t stage1, stage2;      /* preamble gate results (ADSB1090.cpp:782-811) */
    uint8_t p_errors[2];         /* errors counter after slicing, pass 1 / pass 2 */
    uint8_t p_energy_ok[2];      /* delta >= 2550 */
    uint8_t p_msg[2][14];        /* packed bytes before CRC repair */
    uint8_t p_df[2];
    uint8_t p_nbits[2];
    uint8_t phase_applied;       /* pass 2 rescaled the window */
    uint8_t p_crc_state[2];      /* DF11/17 only: 0 bad, 1 ok, 2 repaired */
    int8_t  p_errorbit[2];
    uint8_t p_fixed[2][14];      /* bytes after repair (== p_msg when not repaired) */
    uint32_t p_ap_addr[2];       /* AP xor CRC (address candidate for DF0/4/5/16/20/21/24) */
    uint32_t p_delta[2];         /* energy-gate average (:870-872) */
} oracle1090_probe_t;

typedef struct oracle1090_stats
{
    uint64_t samples, stage1_pass, stage2_pass, sliced, energy_pass, decoded, accepted, retries, phase_applied;
} oracle1090_stats_t;

oracle1090_t* oracle1090_create(void);
void          oracle1090_destroy(oracle1090_t* o);
/* rate_hz == 0: wall clock (reference behaviour).  Otherwise time = t0_ns + stream_sample_index / rate_hz. */
void oracle1090_set_sample_clock(oracle1090_t* o, int64_t t0_ns, uint32_t rate_hz);
/* One RTLSDR::IDataHandler::HandleData call (ADSB1090.cpp:158-175). */
void oracle1090_handle_data(oracle1090_t* o, const uint8_t* data, size_t nbytes, oracle1090_cb cb, void* user);
void oracle1090_get_stats(const oracle1090_t* o, oracle1090_stats_t* out);
size_t oracle1090_aircraft_count(const oracle1090_t* o);

/* building blocks, exposed for unit tests */
const uint16_t* oracle1090_mag_lut(void);                 /* 129*129 entries, index i*129+q (ADSB1090.cpp:131-142) */
void     oracle1090_magnitude(const uint8_t* data, size_t nbytes, uint16_t* m); /* :165-173 */
uint32_t oracle1090_checksum_entry(int idx);             /* ModesChecksumTable[idx], regenerated from the polynomial */
uint32_t oracle1090_checksum(const uint8_t* msg, int bits);      /* :277-291 */
int      oracle1090_fix_single_bit(uint8_t* msg, int bits);      /* :304-332 */
int      oracle1090_msglen_bits(int df);                         /* :295-299 */
void     oracle1090_probe_at(const uint16_t* m, size_t n, size_t j, oracle1090_probe_t* out);
/* every offset j < n-240 that passes both preamble gates, regardless of sequencing; returns the count (may exceed cap) */
size_t   oracle1090_gate_offsets(const uint16_t* m, size_t n, uint32_t* out, size_t cap);
int      oracle1090_cpr_nl(double lat);                          /* :993-1055 */
/* global CPR decode (:1079-1121); returns 0 when the latitude zones disagree (state left untouched) */
int      oracle1090_decode_cpr(double even_lat, double even_lon, double odd_lat, double odd_lon, int use_even, int32_t* lat1e7, int32_t* lon1e7);

/* expected1090.c: the adsb_amd_record_t array (include/adsb_amd.h; 32 bytes each) a scan call over `iq` must produce, from the
 * probes above; buffers spread over `nthreads` threads.  Returns the total count (may exceed cap), (size_t)-1 on failure. */
size_t oracle1090_expected_records(const uint8_t* iq, size_t nbytes, size_t buffer_bytes, void* out, size_t cap, int nthreads);

/* oracle2400.c: the executable specification of the 2.4 MS/s scan mode (no counterpart in the reference; parity unpinned).
 * s = saturated powers, m = magnitudes of one buffer of n samples. */
int    oracle2400_gate(const uint16_t* s, size_t n, size_t j);
int    oracle2400_demod_at(const uint16_t* m, size_t n, size_t j, uint32_t buffer, void* record32_out);
size_t oracle2400_expected_records(const uint8_t* iq, size_t nbytes, size_t buffer_bytes, void* out, size_t cap);

/* UAT978 phase LUT (UAT978.cpp:76-100): 65536 entries indexed by I | Q<<8 */
void oracle978_phase_lut(uint16_t* lut65536);

#ifdef __cplusplus
}
#endif
#endif
