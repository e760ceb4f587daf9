/*
 * adsb_amd.h -- C ABI of libadsb_amd.so: an MI355X (gfx950) IQ -> Mode S frame demodulator that
 * replaces the body of libadsb's 1090 handler while keeping its handler/listener surface.
 *
 * Boundary being replaced (all citations into the reference tree, ankurvdev/libadsb):
 *   RTLSDR::IDataHandler::HandleData(std::span<uint8_t const> const&)      RTLSDR.hpp:39-49
 *   ADSB1090Handler::HandleData -> DetectModeS -> DecodeModesMessage ...     ADSB1090.cpp:158-175, 741-959, 491-675
 *   ADSB::IListener::OnChanged(IAirCraft const&) via TrafficManager          ADSBListener.h:52-60, AircraftImpl.h:49-68
 *   factories ADSB::TryCreateADSB1090Handler / ADSB::test::...               ADSB.h:13-15, 24-26
 *   UAT: extern "C" init_fec / process_buffer / dump_raw_message            UAT978.cpp:9-10, uat2json-wrapper.cpp:7,14
 *
 * The reference has no C ABI for 1090 (the handler is a C++ class); this header is the seam a
 * maintainer binds instead of compiling ADSB1090.cpp -- INTEGRATION.md shows the adapter.
 *
 * Two layers:
 *   adsb_amd_ctx_t       GPU half: u8 IQ -> candidate records (rows a3-a10 of SURVEY.md section 8a)
 *   adsb_amd_resolver_t  host half: records -> accepted frames -> aircraft state -> callback
 *                        (rows a11-a14: ICAO cache, skip-ahead sequencing, field decode, CPR)
 *   adsb_amd_handler_t   both, behind one HandleData-shaped call
 *
 * All functions return 0 on success or a negative ADSB_AMD_E* code; the message is available
 * from *_last_error().  Nothing here falls back to a CPU demodulator: without a usable HIP
 * device adsb_amd_create() fails.
 */
#ifndef ADSB_AMD_H
#define ADSB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADSB_AMD_OK 0
#define ADSB_AMD_ENODEV (-1)   /* no HIP device / HIP runtime error at creation */
#define ADSB_AMD_EINVAL (-2)   /* bad argument (size, alignment, slot) */
#define ADSB_AMD_EHIP (-3)     /* HIP runtime error during a call */
#define ADSB_AMD_ENOSPC (-4)   /* caller's output array too small (n_out holds the needed count) */
#define ADSB_AMD_ESTATE (-5)   /* fetch without submit, etc. */
#define ADSB_AMD_ENOMEM (-6)   /* the record regions an input would need do not fit in free device memory */

/* RTLSDR::BufferLength (RTLSDR.hpp:55): the unit the reference demodulates independently. */
#define ADSB_AMD_REF_BUFFER_BYTES 262144u

/* record.flags */
#define ADSB_AMD_F_PASS2 0x01u      /* produced by the retry slice (reference useCorrection==true) */
#define ADSB_AMD_F_PHASE 0x02u      /* the retry rescaled the window (DetectOutOfPhase != 0, j != 0) */
#define ADSB_AMD_F_NEEDS_ICAO 0x04u /* DF0/4/5/16/20/21/24: valid only if `addr` is in the ICAO cache */

/*
 * One candidate frame emitted by the GPU (32 bytes, little endian).  A record is emitted for every
 * offset that passes the preamble gates, the energy gate and has errors==0, and whose frame the
 * reference could accept: DF11/17 with good or 1-bit-repairable parity (unconditional), or an
 * AP-type DF (conditional on the ICAO cache, ADSB1090.cpp:396-435).  Up to two records per offset
 * (first slice, then the phase-corrected retry).  Records are sorted by (buffer, offset, pass).
 * The sequential rules of ADSB1090.cpp:886-957 (skip-ahead, cache gating) are applied afterwards
 * by the resolver, on the host.
 */
typedef struct adsb_amd_record
{
    uint32_t buffer;   /* index of the reference buffer inside this scan call */
    uint32_t offset;   /* sample index j of the preamble inside that buffer */
    uint32_t addr;     /* DF11/17: bytes 1..3 after repair; AP-type: AP xor parity */
    uint16_t reserved; /* 0 */
    uint8_t  nbits;    /* 56 / 112, from the DF as sliced (before repair) */
    int8_t   errorbit; /* -1, or the repaired bit (FixSingleBitErrors) */
    uint8_t  df;       /* msg[0]>>3 as sliced (before repair) -- Message::msgtype */
    uint8_t  flags;    /* ADSB_AMD_F_* */
    uint8_t  msg[14];  /* message bytes after repair; bytes beyond nbits/8 are 0 (the reference keeps sliced noise there
                          and never reads it) */
} adsb_amd_record_t;

/*
 * The stateless fields of DecodeModesMessage (ADSB1090.cpp:530-672) for one record, computed by the GPU in the ordering pass:
 * what the aircraft update (InteractiveReceiveData, :1124-1175) consumes.  kind: 0 nothing to update, 1 altitude (DF0/4/20, AC13),
 * 2 identification (a, b = the eight callsign characters, first character in the lowest byte of a), 3 airborne position
 * (altitude from AC12, a = raw CPR latitude, b = raw CPR longitude, odd = CPR format flag), 4 airborne velocity (a = speed,
 * b = track in whole degrees, the reference's truncation and wrap).  Parallel to the record array.
 */
typedef struct adsb_amd_decoded
{
    uint8_t  kind, metype, mesub, odd;
    int32_t  altitude;
    uint32_t a, b;
} adsb_amd_decoded_t;

/*
 * A record and its decoded fields in 32 bytes, without the message bytes: everything the aircraft tracker consumes
 * (InteractiveReceiveData, ADSB1090.cpp:1124-1175) and everything of a frame an IListener can observe through IAirCraft (the
 * reference's listener interface never sees message bytes: ADSBListener.h:52-60).  The first 18 bytes are laid out like
 * adsb_amd_record_t's.  This is the hand-over format of the throughput path: 32 instead of 48 bytes per record cross the host link.
 */
typedef struct adsb_amd_packed
{
    uint32_t buffer, offset, addr;
    uint16_t reserved; /* 0 (2.4 MS/s mode: the phase) */
    uint8_t  nbits;
    int8_t   errorbit;
    uint8_t  df, flags; /* as in adsb_amd_record_t */
    uint8_t  kind, odd; /* as in adsb_amd_decoded_t */
    int32_t  altitude;
    uint32_t a, b;
} adsb_amd_packed_t;

/* Accepted frame + aircraft snapshot handed to the callback (mirrors what IListener::OnChanged sees). */
typedef struct adsb_amd_frame
{
    uint64_t offset; /* sample index inside the HandleData buffer (buffer * samples_per_buffer + j) */
    uint8_t  msg[14];
    uint8_t  nbits;
    int8_t   errorbit;
    uint8_t  pass;          /* 1 or 2 */
    uint8_t  phase_applied;
    uint8_t  df;
    uint8_t  reserved;
    uint32_t addr;
} adsb_amd_frame_t;

typedef struct adsb_amd_aircraft
{
    uint32_t addr;
    char     callsign[8]; /* raw, not NUL terminated (AircraftImpl.h:31) */
    int32_t  lat1e7, lon1e7;
    int32_t  altitude;
    uint32_t speed, track;
    int32_t  vert_rate;
    uint32_t squawk;
} adsb_amd_aircraft_t;

typedef void (*adsb_amd_on_changed_fn)(void* user, const adsb_amd_frame_t* frame, const adsb_amd_aircraft_t* aircraft);

const char* adsb_amd_version(void);

/* ---------------------------------------------------------------- GPU half */
typedef struct adsb_amd_ctx adsb_amd_ctx_t;

/* The demodulating part of ADSB1090Handler (constructed at ADSB1090.cpp:144-154): what HandleData (:158-175) -> DetectModeS
 * (:741-959) computes from the samples alone.  device < 0: current HIP device. */
int         adsb_amd_create(adsb_amd_ctx_t** out, int device);
/* The same with an explicit scan mode.  ADSB_AMD_MODE_2000: the reference's demodulator, 2 samples per microsecond
 * (ADSB1090.cpp:148, 741-959) -- what adsb_amd_create gives.  ADSB_AMD_MODE_2400: a second mode for receivers run at 2.4 MS/s, the
 * rate BASELINE.json quotes; libadsb has no demodulator for it (its dump1090 submodule, home of demod_2400.c, is never compiled:
 * CMakeLists.txt:66-83), so this mode is defined by this library (oracle/oracle2400.c is the specification; parity unpinned):
 * five sub-sample phases, Manchester decisions by overlap-weighted magnitude differences, the reference's parity / one-bit
 * repair / AP rules.  Records: `offset` = the sample the frame starts in, `reserved` = the phase (fifths of a sample). */
#define ADSB_AMD_MODE_2000 20
#define ADSB_AMD_MODE_2400 24
int         adsb_amd_create_mode(adsb_amd_ctx_t** out, int device, int mode);
void        adsb_amd_destroy(adsb_amd_ctx_t* ctx);
const char* adsb_amd_last_error(const adsb_amd_ctx_t* ctx); /* ctx may be NULL: creation error */

/*
 * Synchronous scan of host memory (the HandleData case): copies `nbytes` of interleaved u8 I,Q to
 * the device, demodulates, returns the sorted records.  The input is split into independent
 * reference buffers of `buffer_bytes` (0: the whole input is one buffer, as in the reference's
 * TestEmbedded); a trailing partial buffer is ignored, as the reference's replay does
 * (RTLSDR.hpp:429-437).  Each buffer needs >= 480 bytes.
 */
int adsb_amd_scan_1090(adsb_amd_ctx_t* ctx, const uint8_t* iq_host, size_t nbytes, size_t buffer_bytes,
                       adsb_amd_record_t* out, size_t cap, size_t* n_out);

/*
 * Asynchronous scan of device-resident input (no counterpart in the reference, whose HandleData is synchronous; this is the
 * recorded-file / batch form of the same DetectModeS work, ADSB1090.cpp:741-881 + :277-332).  `slot` (0 .. ADSB_AMD_SLOTS - 1) selects one of the
 * context's result buffers so that the device->host copy of one scan overlaps the kernels of the next (two slots do; a third lets a loop keep
 * three scans on the stream; a slot's device arrays are made when it is first used).  `iq_device` must be
 * 16-byte aligned and stay valid until the matching fetch.  `hip_stream` is a hipStream_t
 * (NULL: the context's own stream).  A scan's ordering pass is done by the next scan kernel submitted on the SAME stream (in front of its own
 * work), or by a launch of its own when the scan is fetched first.  A loop may therefore alternate its submits over two streams with three slots
 * in flight: a scan kernel then starts while the one before it drains, and the step costs what the kernel alone costs (profiles/r06_two_streams.txt);
 * the fetch order, and the rule that a slot's next submit follows its fetch (or fetch_packed_begin), stay as they are.
 */
#define ADSB_AMD_SLOTS 3
int adsb_amd_scan_1090_submit(adsb_amd_ctx_t* ctx, const void* iq_device, size_t nbytes, size_t buffer_bytes,
                              void* hip_stream, int slot);
/* Waits for the slot; *records points into context-owned pinned memory, valid until the next submit on that slot. */
int adsb_amd_scan_1090_fetch(adsb_amd_ctx_t* ctx, int slot, const adsb_amd_record_t** records, size_t* n);
/* fetch plus the decoded fields of every record (same order, same lifetime as the record pointer). */
int adsb_amd_scan_1090_fetch_decoded(adsb_amd_ctx_t* ctx, int slot, const adsb_amd_record_t** records, const adsb_amd_decoded_t** decoded,
                                     size_t* n);
/* Which arrays the ordering pass produces, from the next submit on (default ADSB_AMD_OUT_RECORDS | ADSB_AMD_OUT_DECODED).  A fetch of
 * an array the slot's scan did not produce returns ADSB_AMD_ESTATE. */
#define ADSB_AMD_OUT_RECORDS 1u
#define ADSB_AMD_OUT_DECODED 2u
#define ADSB_AMD_OUT_PACKED 4u
int adsb_amd_set_outputs(adsb_amd_ctx_t* ctx, unsigned mask);
/* fetch of the packed form (needs ADSB_AMD_OUT_PACKED): same order, count and lifetime as the records. */
int adsb_amd_scan_1090_fetch_packed(adsb_amd_ctx_t* ctx, int slot, const adsb_amd_packed_t** packed, size_t* n);
/* The same fetch in two halves, for a loop that keeps the GPU fed: _begin waits for the slot's count (*n), starts the copy of the packed records and
 * leaves the slot free for its next adsb_amd_scan_1090_submit; _end waits for that copy and returns the pointer (valid until the slot's next _begin or
 * fetch).  Every _begin has to be followed by its _end before the slot is fetched again. */
int adsb_amd_scan_1090_fetch_packed_begin(adsb_amd_ctx_t* ctx, int slot, size_t* n);
int adsb_amd_scan_1090_fetch_packed_end(adsb_amd_ctx_t* ctx, int slot, const adsb_amd_packed_t** packed, size_t* n);
/* The same wait, but the sorted records are copied into `dst_device` (room for `cap` records: memory of the same GPU, or page-locked /
 * HIP-registered host memory) on `hip_stream` (NULL: an internal stream, and the call returns after the copy has completed): for the
 * hand-over of the sharded recorded-file case -- an RCCL gather of the records from device buffers, or every
 * GPU writing into its segment of node-shared host memory (libadsb_amd/shard.py).  ADSB_AMD_ENOSPC when cap is too
 * small (*n holds the count).  Like every fetch it repeats the scan with more room when the raw-record area overflowed. */
int adsb_amd_scan_1090_fetch_device(adsb_amd_ctx_t* ctx, int slot, void* dst_device, size_t cap, void* hip_stream, size_t* n);
/* the same for the packed form (needs ADSB_AMD_OUT_PACKED; an entry is as large as a record) */
int adsb_amd_scan_1090_fetch_device_packed(adsb_amd_ctx_t* ctx, int slot, void* dst_device, size_t cap, void* hip_stream, size_t* n);
/* Device time of the last completed scan on `slot`: the demodulation kernel alone (HIP events around its launch, on the stream it was
 * submitted to), and scan start to record count on the host.  ADSB_AMD_ESTATE when that scan was not timed (adsb_amd_set_timing). */
int adsb_amd_scan_1090_timing(adsb_amd_ctx_t* ctx, int slot, float* scan_kernel_ms, float* total_ms);
/* Which scans get the two timing events around the demodulation kernel: every `every`-th submit of the context (1, the default:
 * all; 0: none).  An event costs the scan's stream 3-5 us, two of them 3 % of a 1 GiB scan: a caller that pipelines scans back to back
 * and wants the kernel time as a running figure samples it. */
int adsb_amd_set_timing(adsb_amd_ctx_t* ctx, unsigned every);

/* Parity helper: magnitudes exactly as ADSB1090.cpp:165-173 computes them (n = nbytes/2 values). */
int adsb_amd_magnitude_1090(adsb_amd_ctx_t* ctx, const uint8_t* iq_host, size_t nbytes, uint16_t* mag_out);
/* Parity helpers: the field decoder of the ordering pass run on the device over arbitrary records (only msg and df are read),
 * and the host build of the same function (one record; needs no GPU). */
int  adsb_amd_decode_1090(adsb_amd_ctx_t* ctx, const adsb_amd_record_t* records_host, size_t n, adsb_amd_decoded_t* out_host);
void adsb_amd_decode_record_host(const adsb_amd_record_t* record, adsb_amd_decoded_t* out);

/* ---------------------------------------------------------------- host half */
/* The order-dependent rest of ADSB1090Handler: skip-ahead and retry order (ADSB1090.cpp:886-957), ICAO cache and BruteForceAp
 * (:195-207, 396-435), DecodeModesMessage (:491-675), UseModesMessage / InteractiveReceiveData / DecodeCpr (:968-1175). */
typedef struct adsb_amd_resolver adsb_amd_resolver_t;

adsb_amd_resolver_t* adsb_amd_resolver_create(void);
void                 adsb_amd_resolver_destroy(adsb_amd_resolver_t* r);
/* rate_hz == 0 (default): wall clock, as the reference.  Otherwise time = t0_ns + stream sample index / rate_hz. */
void adsb_amd_resolver_set_sample_clock(adsb_amd_resolver_t* r, int64_t t0_ns, uint32_t rate_hz);
/* Samples per microsecond x 10 of the records fed (ADSB_AMD_MODE_*): how many samples an accepted frame hides (ADSB1090.cpp:929-934
 * skips (8 + bits) * 2; at 2.4 MS/s the same frame spans (8 + bits) * 2.4 samples).  Default 20. */
void adsb_amd_resolver_set_mode(adsb_amd_resolver_t* r, int mode);
/*
 * Applies the reference's sequential rules to sorted records of one scan call and fires the
 * callback once per accepted frame, in sample order.  `samples_per_buffer` and `nbuffers` describe
 * the scan call (they advance the stream position used by the sample clock).
 * Returns the number of accepted frames (>= 0) or a negative error.
 */
long adsb_amd_resolver_feed(adsb_amd_resolver_t* r, const adsb_amd_record_t* records, size_t n, size_t samples_per_buffer,
                            size_t nbuffers, adsb_amd_on_changed_fn cb, void* user);
/* The same with the GPU's decoded fields (adsb_amd_scan_1090_fetch_decoded): the host does no field decoding at all. */
long adsb_amd_resolver_feed_decoded(adsb_amd_resolver_t* r, const adsb_amd_record_t* records, const adsb_amd_decoded_t* decoded, size_t n,
                                    size_t samples_per_buffer, size_t nbuffers, adsb_amd_on_changed_fn cb, void* user);
/* The same from the packed form.  The frames handed to the callback carry every field but the message bytes (msg is all zero). */
long adsb_amd_resolver_feed_packed(adsb_amd_resolver_t* r, const adsb_amd_packed_t* packed, size_t n, size_t samples_per_buffer, size_t nbuffers,
                                   adsb_amd_on_changed_fn cb, void* user);
size_t adsb_amd_resolver_aircraft_count(const adsb_amd_resolver_t* r);
/* Parity helpers (host only): CprNlFunction (ADSB1090.cpp:993-1055) and the global airborne decode (:1079-1121) as the resolver
 * computes them; the four arguments of adsb_amd_cpr_global are the raw 17-bit CPR values (whole numbers; a fraction is dropped); it returns 0
 * when the two latitudes fall into different zones (state untouched). */
int adsb_amd_cpr_nl(double lat);
int adsb_amd_cpr_global(double even_lat, double even_lon, double odd_lat, double odd_lon, int use_even, int32_t* lat1e7, int32_t* lon1e7);
/* The batch form the resolver runs on the even/odd pairs of a batch of frames (four pairs per step, AVX2): ok[i] = 1 and lat1e7[i],
 * lon1e7[i] written where pair i decodes, ok[i] = 0 and the outputs untouched where adsb_amd_cpr_global returns 0.  Same results pair by pair. */
void adsb_amd_cpr_global_batch(size_t n, const int32_t* even_lat, const int32_t* even_lon, const int32_t* odd_lat, const int32_t* odd_lon,
                               const uint8_t* use_even, int32_t* lat1e7, int32_t* lon1e7, uint8_t* ok);
/* An adsb_amd_on_changed_fn that only counts: *(uint64_t*)user += 1 per call (IListener::OnChanged stand-in for rate runs). */
void adsb_amd_count_callback(void* user, const adsb_amd_frame_t* frame, const adsb_amd_aircraft_t* aircraft);

/* ---------------------------------------------------------------- both: one HandleData-shaped call */
/* ADSB1090Handler as a whole (ADSB1090.cpp:99-244): create/destroy = ctor/dtor (:144-154), handle_data = HandleData (:158-175). */
typedef struct adsb_amd_handler adsb_amd_handler_t;

int         adsb_amd_handler_create(adsb_amd_handler_t** out, int device);
int         adsb_amd_handler_create_mode(adsb_amd_handler_t** out, int device, int mode); /* ADSB_AMD_MODE_*; see adsb_amd_create_mode */
void        adsb_amd_handler_destroy(adsb_amd_handler_t* h);
const char* adsb_amd_handler_last_error(const adsb_amd_handler_t* h);
void        adsb_amd_handler_set_sample_clock(adsb_amd_handler_t* h, int64_t t0_ns, uint32_t rate_hz);
/* want_frames = 0: the listener looks at aircraft only (what libadsb's IListener can see), so the handler moves records in the packed
 * form and the frames passed to the callback have no message bytes (msg all zero).  Default 1. */
void        adsb_amd_handler_set_frames(adsb_amd_handler_t* h, int want_frames);
/* RTLSDR::IDataHandler::HandleData: synchronous; callbacks fire before it returns, in sample order.
 * buffer_bytes as in adsb_amd_scan_1090 (0 = the reference's behaviour: one call, one buffer). */
long adsb_amd_handler_handle_data(adsb_amd_handler_t* h, const uint8_t* iq_host, size_t nbytes, size_t buffer_bytes,
                                  adsb_amd_on_changed_fn cb, void* user);
/* Page-locked host memory for the transport's ring slots (RTLSDR.hpp:564-570 keeps BufferCount x BufferLength bytes):
 * HandleData on a buffer that lives in it uploads by DMA straight from the slot.  Optional; any host pointer works. */
int  adsb_amd_host_alloc(void** out, size_t nbytes);
void adsb_amd_host_free(void* p);

/* The recorded-file job over several GPUs of one node (BASELINE configs[3]; nothing the reference has: its one receiver thread is
 * RTLSDR.hpp:470-473) hands each rank's sorted records to the resolving rank through node-shared host memory; a step's 32-byte header
 * {count, first buffer, rank, step} goes through a control page that several PROCESSES map.  These are the only accesses to that page:
 * C11 release / acquire on naturally aligned 64-bit words, so that the ordering "a header the root can see implies records it can see"
 * does not rest on an interpreter's call boundaries or on one architecture's store order.
 *   post_header   three relaxed stores (count, first buffer, rank), then `step` with release into slot[3]
 *   read_header   acquire load of slot[3]; 0 when it is below min_step, else the four words in out[] and 1
 *   store_release / load_acquire: the credit word (last step the resolving rank has finished reading) */
void    adsb_amd_shm_post_header(int64_t* slot, int64_t count, int64_t first_buffer, int64_t rank, int64_t step);
int     adsb_amd_shm_read_header(const int64_t* slot, int64_t min_step, int64_t* out4);
void    adsb_amd_shm_store_release(int64_t* word, int64_t value);
int64_t adsb_amd_shm_load_acquire(const int64_t* word);

/* Recorded-file replay, one pass: what RTLSDR::TestDataReadLoop (RTLSDR.hpp:419-442) feeds a handler -- whole 262144-byte
 * buffers in file order, each demodulated on its own, a trailing partial buffer never delivered -- over buffers
 * [first_buffer, first_buffer + max_buffers) of the file (ranks of a multi-GPU job take disjoint ranges; the resolver state
 * of this handler carries across the whole range; a frame's `offset` counts samples from the first buffer of the pass, as if the whole
 * range had been one HandleData call).  Returns the number of accepted frames or a negative error. */
long adsb_amd_handler_replay_file(adsb_amd_handler_t* h, const char* path, size_t first_buffer, size_t max_buffers, adsb_amd_on_changed_fn cb,
                                  void* user);

/* ---------------------------------------------------------------- sample transport (stand-alone stand-in for RTLSDR's ring)
 * What the reference's RTLSDR class does for a handler, without librtlsdr (RTLSDR.hpp:55-56, 396-442, 493-539, 564-570): a ring of
 * 16 slots of 262144 bytes (page-locked when a GPU runtime is present) between one producer and one consumer thread.  The producer
 * is the replay of a recorded file (replay_path: whole 262144-byte reads in file order, re-opened at its end when loop != 0, a
 * trailing partial read never delivered) or the caller of adsb_amd_transport_push (RTLSDR::OnDataAvailable: the USB callback of
 * a host that owns the receiver; blocks while the ring is full).  The consumer thread hands one slot at a time to `sink`. */
typedef struct adsb_amd_transport adsb_amd_transport_t;
typedef void (*adsb_amd_buffer_fn)(void* user, const uint8_t* data, size_t nbytes);
int  adsb_amd_transport_create(adsb_amd_transport_t** out, const char* replay_path /* NULL: push mode */, int loop);
void adsb_amd_transport_destroy(adsb_amd_transport_t* t);
int  adsb_amd_transport_start(adsb_amd_transport_t* t, adsb_amd_buffer_fn sink, void* user); /* EINVAL when the file cannot be opened */
int  adsb_amd_transport_stop(adsb_amd_transport_t* t);                                       /* joins both threads */
int  adsb_amd_transport_push(adsb_amd_transport_t* t, const uint8_t* data, size_t nbytes);    /* EINVAL unless nbytes % 262144 == 0 */
int  adsb_amd_transport_stats(const adsb_amd_transport_t* t, uint64_t* delivered, int* producer_done, int* page_locked);
/* The reference's replay mode end to end in native code: the file through the ring, one HandleData per slot on the consumer
 * thread (callbacks fire there, in order), until the file has been delivered once.  Returns the accepted frames or an error;
 * *buffers / *seconds (optional) receive the slots delivered and the wall time. */
long adsb_amd_handler_run_replay(adsb_amd_handler_t* h, const char* path, adsb_amd_on_changed_fn cb, void* user, uint64_t* buffers, double* seconds);

/* =====================================================================================================================
 * UAT 978.  Boundary replaced:
 *   UAT978Handler::HandleData(std::span<uint8_t const>)                     UAT978.cpp:43-60   -> adsb_amd_uat_handle_data
 *   extern "C" void init_fec(); int process_buffer(uint16_t const*, int, uint64_t)   UAT978.cpp:9-10 -> same names, below
 *   extern "C" void dump_raw_message(char, uint8_t*, int, int)               uat2json-wrapper.cpp:14 (the up-call, host's)
 * The arithmetic behind process_buffer is the un-vendored dump978 module: this side restates the published
 * legacy algorithm; parity unpinned (checked against oracle/oracle978.c only).
 * Division of work: phases (LUT), sign of the phase difference, every 18-bit sync match, the 36-bit sync re-check, the
 * frame slicing and the Reed-Solomon decode run on the GPU; only the order-dependent scan-loop rules run on the host.
 * ===================================================================================================================*/
typedef struct adsb_amd_uat adsb_amd_uat_t;

/* dump_raw_message(updown, data, len, rs_errors) plus the stream sample index of the frame's first sync sample;
 * updown '-' = ADS-B downlink (len 18 or 34), '+' = ground uplink (len 432) */
typedef void (*adsb_amd_uat_frame_fn)(void* user, char updown, const uint8_t* data, int len, int rs_errors, uint64_t sample_index);
typedef void (*adsb_amd_dump_raw_message_fn)(char updown, uint8_t* data, int len, int rs_errors);

int         adsb_amd_uat_create(adsb_amd_uat_t** out, int device);
void        adsb_amd_uat_destroy(adsb_amd_uat_t* u);
const char* adsb_amd_uat_last_error(const adsb_amd_uat_t* u); /* u == NULL: error of the last failed create on this thread */

/* UAT978Handler::HandleData: u8 IQ pairs in, staged 65 536 phases at a time exactly as UAT978.cpp:48-59 does, frames out
 * through cb in stream order.  The reference passes the unconsumed tail's ENTRY count as memmove's BYTE count (:57), so only
 * half of the tail really carries over; that is reproduced by default.  set_carry_full(1) carries the whole tail. */
int adsb_amd_uat_handle_data(adsb_amd_uat_t* u, const uint8_t* iq_host, size_t nbytes, adsb_amd_uat_frame_fn cb, void* user);
int adsb_amd_uat_set_carry_full(adsb_amd_uat_t* u, int full);
/* Which frames the dump978 scan loop takes is decided on the device (a successor function over the ordered matches, resolved by
 * pointer jumping).  on != 0 makes the host walk that loop over the device's records instead, as rounds 1-2 did (also:
 * ADSB_AMD_UAT_HOST_LOOP=1 in the environment at create time); the frames are the same, tests hold the two against each other. */
int adsb_amd_uat_set_host_loop(adsb_amd_uat_t* u, int on);
/* Frames the loop reaches only through stale register bits right after a jump are demodulated on the device as well, into a side
 * array of 4096 entries per call (about 19 are used per GiB of frame-dense stream).  A call that needs more is walked by the host
 * loop instead, with the same result.  This lowers the number of entries (<= 4096); tests use it to reach that path. */
int adsb_amd_uat_set_extra_capacity(adsb_amd_uat_t* u, uint32_t entries);
int adsb_amd_uat_stream_state(const adsb_amd_uat_t* u, uint64_t* offset, size_t* used); /* UAT978Handler::offset / used */

/* process_buffer over one buffer of any length < 2^31 samples (the reference passes <= 65 536): phases from the host, or u8
 * IQ from the host / already in HBM (on_device != 0, 16-byte aligned).  *consumed = samples the caller may drop (negative
 * when the buffer is shorter than one maximum frame, as in dump978). */
int adsb_amd_uat_process_phases(adsb_amd_uat_t* u, const uint16_t* phi_host, uint64_t len, uint64_t offset, adsb_amd_uat_frame_fn cb, void* user,
                                int64_t* consumed);
int adsb_amd_uat_process_iq(adsb_amd_uat_t* u, const void* iq, uint64_t nsamples, int on_device, uint64_t offset, adsb_amd_uat_frame_fn cb,
                            void* user, int64_t* consumed);
/* The same in two halves, for streams already in HBM: submit starts the GPU half (match search, ordering, per-match
 * demodulation and Reed-Solomon, records to the host) on a worker thread and returns; collect waits for the oldest submitted call and
 * runs the scan loop with its up-calls on the caller's thread.  Three calls may be in flight (each on its own stream and buffers), so
 * the GPU halves of calls k + 1 and k + 2 overlap the scan loop of call k; a fourth submit returns ADSB_AMD_ESTATE.  The input of a call must stay
 * valid until it is collected.  Results are identical to adsb_amd_uat_process_iq call by call.  While submitted calls are uncollected the
 * synchronous entry points of the same handle (handle_data, process_phases, process_iq) return ADSB_AMD_ESTATE: they share its buffers. */
int adsb_amd_uat_submit_iq(adsb_amd_uat_t* u, const void* iq_device, uint64_t nsamples, uint64_t offset);
int adsb_amd_uat_max_in_flight(void);
/* One process_buffer over a stream that is cut over several GPUs ("shard with a halo").  A part's window holds
 * its own samples [own_begin, own_end), at least 64 samples before them (unless the stream begins there) and, unless it is the stream's
 * last part, three maximum frames + 64 samples (3 * 2 * (36 + 4416) + 64) behind them.  part_scan does everything that does not depend
 * on the parts before (match search, demodulation, Reed-Solomon: the heavy half, all parts at once); part_finish takes the bit at which
 * the scan loop stands when it comes into the part -- 0 for the first part, the previous part's *exit_bit otherwise, rebased to this
 * window (bits = samples / 2) --, makes the up-calls for the frames whose start bit is the part's own, and returns the bit at which the
 * loop leaves it.  The frames of the parts, in part order, and the last part's *consumed (+ its window's first sample) are what
 * adsb_amd_uat_process_iq returns for the whole stream.  ADSB_AMD_ENOSPC: a chain of frames behind stale register bits left the window. */
int adsb_amd_uat_part_scan(adsb_amd_uat_t* u, const void* iq_device, uint64_t nsamples);
int adsb_amd_uat_part_finish(adsb_amd_uat_t* u, int64_t own_begin_sample, int64_t own_end_sample, int64_t entry_bit, int last, uint64_t offset,
                             adsb_amd_uat_frame_fn cb, void* user, int64_t* exit_bit, int64_t* consumed);
int adsb_amd_uat_collect(adsb_amd_uat_t* u, adsb_amd_uat_frame_fn cb, void* user, int64_t* consumed);
/* Parity helpers for the CPU tests: the scan loop's filter for the 17 steps after a jump (bit t set = step t can still fire, given a
 * register's 18 old bits and the bits that enter it, both in stream order), and the 18-bit check words in the same order. */
uint32_t adsb_amd_uat_possible_steps(uint32_t old_register, uint32_t fresh_bits);
uint32_t adsb_amd_uat_check_word(int uplink);
/* device time of the last process call (sign+match kernels, demod kernel) and running totals of 18-bit matches and of
 * positions outside the match list that the loop reached through stale register bits (demodulated by the wave of the frame
 * before them; with the host loop: asked for by the host one by one) */
int adsb_amd_uat_timing(const adsb_amd_uat_t* u, float* scan_ms, float* demod_ms, uint64_t* candidates, uint64_t* extra_lookups);
/* host wall time of the last process call, by stage: launch..match count on the host, ordering + demod + decision kernels..records
 * on the host, [sort_ms: device time of the decision kernels alone], the walk over the frames taken (including up-calls) */
int adsb_amd_uat_host_timing(const adsb_amd_uat_t* u, float* match_ms, float* demod_ms, float* sort_ms, float* loop_ms);
int adsb_amd_uat_phase_lut(const adsb_amd_uat_t* u, uint16_t* lut65536);  /* InitATan2Table, UAT978.cpp:76-100 */
int adsb_amd_uat_rs_decode(int kind, uint8_t* codeword); /* 0 RS(30,18), 1 RS(48,34), 2 RS(92,72); in place; corrected count or -1 */
/* the same on the GPU, with the decoder the demod kernel uses: `count` packed code words in place, one result each */
int adsb_amd_uat_rs_decode_device(adsb_amd_uat_t* u, int kind, uint8_t* codewords, int count, int* results);

/* The reference's own seam, for a host that links this library instead of dump978's legacy objects.  init_fec() creates a
 * process-wide context on device $ADSB_AMD_DEVICE (default 0) and aborts with a message when there is no usable GPU;
 * process_buffer() reports frames through dump_raw_message (weakly bound: the host's definition, uat2json-wrapper.cpp:14)
 * or through the function registered here. */
void init_fec(void);
int  process_buffer(const uint16_t* phi, int len, uint64_t offset);
void adsb_amd_uat_set_dump_raw_message(adsb_amd_dump_raw_message_fn fn);


#ifdef __cplusplus
}
#endif
#endif /* ADSB_AMD_H */
