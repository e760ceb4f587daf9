// capi.cpp -- C ABI of libadsb_amd.so (include/adsb_amd.h): GPU context, two-slot scan pipeline, handler.
// Host-side only; the kernels live in scan1090.hip.  There is deliberately no CPU demodulation path here:
// every entry point that needs the device fails with ADSB_AMD_ENODEV / ADSB_AMD_EHIP when HIP is unusable.
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cerrno>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "adsb_amd.h"
#include "decode1090.h"
#include "diag.hip.h"
#include "resolver1090.hpp"
#include "scan1090.h"
#include "transport.hpp"

using namespace adsb_amd;

namespace
{
thread_local std::string g_create_error;

struct Slot
{
    uint32_t*          counts   = nullptr; // the chunk directory: records kept per chunk
    uint32_t*          block_sums = nullptr; // two arrays of one padded entry per 256 chunks: the scan adds into one, the ordering pass zeroes the other
    size_t             sums_words = 0;       // words per array
    int                sums_phase = 0;       // which array the next scan uses
    adsb_amd_record_t* regions  = nullptr; // total_chunks * cap
    adsb_amd_record_t* dense    = nullptr;
    adsb_amd_decoded_t* decoded = nullptr; // parallel to dense
    adsb_amd_packed_t*  packed  = nullptr; // parallel to dense (allocated when the packed form was asked for)
    unsigned            produced = 0;      // ADSB_AMD_OUT_* of the slot's last scan: the context's mask when it was SUBMITTED (a repeat after an overflow keeps it)
    uint32_t*          total_d  = nullptr; // device {total, overflow}
    uint32_t*          work_d   = nullptr; // device: one chunk counter per XCD (scan1090_kernel), zero between scans
    unsigned long long* total_h = nullptr; // page-locked: count | (stamp << 1 | overflow) << 32, stored by the ordering pass
    unsigned long long* total_h_dev = nullptr; // the device's address of it
    adsb_amd_record_t* host     = nullptr; // pinned result
    adsb_amd_decoded_t* host_dec = nullptr; // pinned, parallel to host (same capacity)
    adsb_amd_packed_t*  host_packed = nullptr; // pinned, same capacity
    bool               dec_valid = false;  // host_dec holds the last fetch's decoded fields
    size_t             host_cap = 0;       // records
    size_t             chunks_cap = 0, cap_per_chunk = 0;
    size_t             cap_hint = 0;       // region size another slot had to grow to: this slot's next scan starts with it
    hipEvent_t         ev_scan0 = nullptr, ev_scan1 = nullptr, ev_order = nullptr;
    uint32_t           seq = 0;            // launches of this slot: the stamp the ordering pass writes beside the count
    bool               pending = false, timed = false;
    // Round 6: the slot's ordering pass is not launched with its scan.  The NEXT scan kernel on the same stream (the other slot's) does it in front of
    // its own work (gather1090.hip.h) -- `gather_attached` --, or, when the slot is waited for before any such kernel came, a launch of its own does.
    bool               gather_launched = false, gather_attached = false;
    bool               copying = false;  // adsb_amd_scan_1090_fetch_packed_begin has put the slot's records on the copy stream; _end has not been called
    size_t             copy_n  = 0;      // their number
    uint32_t*          next_sums = nullptr; // the sum array this scan's ordering pass zeroes (the slot's other one)
    unsigned long long submit_no = 0;       // adsb_amd_ctx::submit_no when this scan was enqueued
    bool               events = false;     // this scan has the two timing events around its kernel
    // the submitted job (needed again when a chunk region overflows and the scan is repeated with a larger cap)
    ScanArgs    args{};
    hipStream_t stream   = nullptr;
    size_t      nrecords = 0;
    bool        overflow = false;
    float       scan_ms = 0.f, total_ms = 0.f;
};
} // namespace

constexpr int kSlots = 3; // result slots of a context (adsb_amd.h: ADSB_AMD_SLOTS)
static_assert(kSlots == ADSB_AMD_SLOTS, "adsb_amd.h");

struct adsb_amd_ctx
{
    int         device = 0;
    int         mode   = ADSB_AMD_MODE_2000;
    uint32_t    nxcd = 8, ncu = 256; // topology of the device, read once at create
    hipStream_t stream = nullptr, copy_stream = nullptr;
    uint32_t*   crc_tab = nullptr;
    uint16_t*   lut978  = nullptr;
    uint8_t*    staging = nullptr; // device copy of host input
    size_t      staging_cap = 0;
    Slot        slot[kSlots];
    unsigned long long submit_no = 0;     // submits so far (a slot notes its own: the oldest scan waiting for its ordering pass is the one a new scan kernel takes on)
    unsigned    outputs = ADSB_AMD_OUT_RECORDS | ADSB_AMD_OUT_DECODED; // arrays the ordering pass produces (adsb_amd_set_outputs)
    unsigned    timing_every = 1, submits = 0; // adsb_amd_set_timing
    unsigned long long* stamps_d = nullptr; // measurement builds (diag.hip.h): four clock values per wave and launch, a ring of kStampSteps launches
    unsigned            stamp_no = 0;
    std::string error;
};

namespace
{
#define HIP_TRY(ctx, expr)                                                                              \
    do                                                                                                  \
    {                                                                                                   \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
        {                                                                                               \
            (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(_e);                           \
            return ADSB_AMD_EHIP;                                                                       \
        }                                                                                               \
    } while (0)

int fail(adsb_amd_ctx* c, int code, const char* msg)
{
    c->error = msg;
    return code;
}

constexpr uint32_t kDefaultCap = 32;
constexpr unsigned kStampSteps = 128, kStampGroups = 8192; // measurement builds: launches kept (a ring), workgroups per launch (scan_common.hip.h)

// How the record count of a scan reaches the host: the ordering pass's last workgroup writes {count, overflow flag, launch stamp} as ONE
// 8-byte system-scope store into page-locked host memory and the host polls the stamp; nothing but the two kernels is on the scan's
// stream.  (Round 3 copied the two words and recorded an event on the scan's stream: 11 us of that stream per step.  A copy + event on a
// control stream behind the pass's event was measured too: the runtime's 8-byte copy is a kernel, beside the 2.4 MS/s scan it got onto the
// chip 0.1-0.2 ms late and the GPU idled 75 us per step -- and a stream told to wait for an event that rides on a dispatch and is re-used
// every launch did not reliably wait.  Both forms are gone; profiles/r04_step_timeline.txt has their figures.)

void free_slot(Slot& s)
{
    if (s.counts) (void)hipFree(s.counts);
    if (s.block_sums) (void)hipFree(s.block_sums);
    if (s.regions) (void)hipFree(s.regions);
    if (s.dense) (void)hipFree(s.dense);
    if (s.decoded) (void)hipFree(s.decoded);
    if (s.packed) (void)hipFree(s.packed);
    s.packed = nullptr;
    s.counts = s.block_sums = nullptr;
    s.regions = s.dense = nullptr;
    s.decoded = nullptr;
    s.chunks_cap = s.cap_per_chunk = 0;
}

int ensure_slot(adsb_amd_ctx* c, Slot& s, size_t chunks, size_t cap, unsigned mask)
{
    if (chunks <= s.chunks_cap && cap == s.cap_per_chunk)
    {
        if ((mask & ADSB_AMD_OUT_PACKED) && !s.packed) HIP_TRY(c, hipMalloc(&s.packed, s.chunks_cap * s.cap_per_chunk * sizeof(adsb_amd_packed_t)));
        return ADSB_AMD_OK;
    }
    free_slot(s);
    size_t nch = chunks ? chunks : 1;
    HIP_TRY(c, hipMalloc(&s.counts, nch * sizeof(uint32_t)));
    s.sums_words = sums_entries(nch) * kSumStride;
    s.sums_phase = 0;
    HIP_TRY(c, hipMalloc(&s.block_sums, 2 * s.sums_words * sizeof(uint32_t)));
    HIP_TRY(c, hipMemset(s.block_sums, 0, 2 * s.sums_words * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc(&s.regions, nch * cap * sizeof(adsb_amd_record_t)));
    HIP_TRY(c, hipMalloc(&s.dense, nch * cap * sizeof(adsb_amd_record_t)));
    HIP_TRY(c, hipMalloc(&s.decoded, nch * cap * sizeof(adsb_amd_decoded_t)));
    if (mask & ADSB_AMD_OUT_PACKED) HIP_TRY(c, hipMalloc(&s.packed, nch * cap * sizeof(adsb_amd_packed_t)));
    s.chunks_cap    = nch;
    s.cap_per_chunk = cap;
    return ADSB_AMD_OK;
}

int ensure_host(adsb_amd_ctx* c, Slot& s, size_t nrec)
{
    if (nrec <= s.host_cap) return ADSB_AMD_OK;
    if (s.host) (void)hipHostFree(s.host);
    if (s.host_dec) (void)hipHostFree(s.host_dec);
    if (s.host_packed) (void)hipHostFree(s.host_packed);
    s.host        = nullptr;
    s.host_dec    = nullptr;
    s.host_packed = nullptr;
    s.host_cap    = 0;
    size_t want   = nrec + nrec / 4 + 1024;
    HIP_TRY(c, hipHostMalloc(&s.host, want * sizeof(adsb_amd_record_t), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc(&s.host_dec, want * sizeof(adsb_amd_decoded_t), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc(&s.host_packed, want * sizeof(adsb_amd_packed_t), hipHostMallocDefault));
    s.host_cap = want;
    return ADSB_AMD_OK;
}

// Describe one scan call.  buffer_bytes == 0: one buffer spanning the whole input.
int make_args(adsb_amd_ctx* c, const void* iq_device, size_t nbytes, size_t buffer_bytes, ScanArgs* a)
{
    if (((uintptr_t)iq_device & 15u) != 0) return fail(c, ADSB_AMD_EINVAL, "iq_device must be 16-byte aligned");
    size_t bb   = buffer_bytes ? buffer_bytes : (nbytes & ~(size_t)1);
    size_t nbuf = bb ? nbytes / bb : 0;
    if (buffer_bytes == 0 && nbytes < 480) nbuf = 0; // the reference underflows its loop bound below 480 bytes; we decline
    if (bb & 1u) return fail(c, ADSB_AMD_EINVAL, "buffer_bytes must be even");
    if (nbuf > 1 && (bb & 15u) != 0) return fail(c, ADSB_AMD_EINVAL, "buffer_bytes must be a multiple of 16 when the input holds several buffers");
    if (nbuf > 0 && bb < 480) return fail(c, ADSB_AMD_EINVAL, "each buffer needs at least 480 bytes (240 samples)");
    if (bb / 2 > 0xFFFFFF00ull) return fail(c, ADSB_AMD_EINVAL, "buffer too large (sample index must fit 32 bits)");
    std::memset(a, 0, sizeof(*a));
    a->iq             = static_cast<const uint8_t*>(iq_device);
    a->buf_stride     = bb;
    a->buf_samples    = (uint32_t)(bb / 2);
    a->nbuf           = (uint32_t)nbuf;
    a->chunks_per_buf = c->mode == ADSB_AMD_MODE_2400 ? chunks_per_buffer_2400(a->buf_samples) : chunks_per_buffer(a->buf_samples);
    a->cpb_magic      = a->chunks_per_buf > 1 ? (uint32_t)((1ull << 32) / a->chunks_per_buf) : 0xFFFFFFFFu;
    uint64_t total    = (uint64_t)a->chunks_per_buf * nbuf;
    if (total > 0x7FFFFFFFull) return fail(c, ADSB_AMD_EINVAL, "input too large for one scan call");
    a->total_chunks = (uint32_t)total;
    a->main_chunks  = a->total_chunks;
    // groups of 16 neighbouring chunks per XCD once every work counter gets several of them; chunk by chunk for small inputs (a live
    // 262144-byte buffer is 32 chunks: they must spread over 32 waves, not queue up behind two)
    a->group_log2 = total >= 16ull * 8 * c->nxcd * kSubRanges ? 4u : 0u;
    a->crc_tab      = c->crc_tab;
    a->nxcd         = c->nxcd;
    a->ncu          = c->ncu;
    // The last sixteenth of a large input (a quarter: 0.2195 ms, an eighth: 0.2131, a sixteenth: 0.2116) is not pre-assigned to an XCD (take_next, scan_common.hip.h): whole rounds of groups over all counters stay in
    // the main part, so every counter's share is the same number of whole groups.  A pool only where every wave is sure to reach it: the main part
    // gives every wave at least four work items (its two fixed ones, then tickets).
    if (a->group_log2)
    {
        const uint32_t round = (c->nxcd * kSubRanges) << a->group_log2;
        const uint32_t main  = (a->total_chunks - a->total_chunks / 16u) / round * round;
        if ((uint64_t)main >= 4ull * scan_grid(*a)) a->main_chunks = main;
    }
    return ADSB_AMD_OK;
}

// what the ordering pass of the slot's current scan needs (scan1090.h)
GatherArgs gather_args(const Slot& s)
{
    GatherArgs g;
    g.chunk_records = s.args.chunk_records, g.chunk_dir = s.args.chunk_dir, g.block_sums = s.args.block_sums;
    g.nchunks = s.args.total_chunks, g.nblocks = (s.args.total_chunks + kOrderChunks - 1u) / kOrderChunks;
    g.cap = s.args.cap, g.chunks_per_buf = s.args.chunks_per_buf ? s.args.chunks_per_buf : 1u;
    g.dense   = (s.produced & ADSB_AMD_OUT_RECORDS) ? s.dense : nullptr;
    g.decoded = (s.produced & ADSB_AMD_OUT_DECODED) ? s.decoded : nullptr;
    g.packed  = (s.produced & ADSB_AMD_OUT_PACKED) ? s.packed : nullptr;
    g.state   = s.total_d;
    g.next_block_sums = s.next_sums, g.next_entries = (uint32_t)(s.sums_words / kSumStride), g.work_counters = s.work_d;
    g.host_word = s.total_h_dev, g.stamp = s.seq, g.stamps = s.args.stamps;
    return g;
}

int enqueue(adsb_amd_ctx* c, Slot& s)
{
    s.args.chunk_records = s.regions;
    s.args.chunk_dir     = s.counts;
    s.args.cap           = (uint32_t)s.cap_per_chunk;
    s.args.work_counters = s.work_d;
    s.args.block_sums    = s.block_sums + (size_t)s.sums_phase * s.sums_words;
    uint32_t* next_sums  = s.block_sums + (size_t)(s.sums_phase ^ 1) * s.sums_words;
    if (s.args.total_chunks) s.sums_phase ^= 1; // an empty input launches nothing: the arrays keep their roles
    // Everything on the caller's stream.  (Running the ordering pass on a second stream beside the next scan was measured
    // slower, twice: with the fixed-stride scan its workgroups delayed persistent waves and the scan grew a tail; with
    // the work counters, and even with one wave slot per CU left free, the step went from 0.31 to 0.50 ms -- the small
    // kernels do not get onto the chip while 4096 persistent workgroups are being placed.  Round 3, a third time, with what could
    // have kept it off a CU removed: the pass rewritten without LDS (a scan's sixteen waves hold all of a CU's) and held to 64 VGPRs
    // (the scan's waves leave that many per SIMD) -- 0.263 -> 0.313 and 0.280 -> 0.333 ms per step.)
    s.args.stamps = c->stamps_d ? c->stamps_d + (size_t)4 * kStampGroups * (c->stamp_no++ % kStampSteps) : nullptr;
    s.next_sums   = next_sums;
    s.seq++;
    s.events = c->timing_every != 0 && (c->submits++ % c->timing_every) == 0;
    // the two timing events ride on the kernel's dispatch (scan1090.h); an empty input launches nothing and is not timed
    s.events = s.events && s.args.total_chunks != 0;
    // The oldest scan of another slot that is still waiting for its ordering pass on this very stream: this kernel's waves do that pass, in front of
    // their own work (round 6).
    Slot* op = nullptr;
    for (Slot& x : c->slot)
        if (&x != &s && x.pending && !x.gather_launched && x.stream == s.stream && x.args.total_chunks != 0 && (!op || x.submit_no < op->submit_no)) op = &x;
    const bool        attach = op != nullptr && s.args.total_chunks != 0;
    Slot&             o      = attach ? *op : s;
    const GatherArgs  ga     = attach ? gather_args(o) : GatherArgs{};
    const GatherArgs* gp     = attach ? &ga : nullptr;
    s.submit_no              = ++c->submit_no;
    if (c->mode == ADSB_AMD_MODE_2400) HIP_TRY(c, launch_scan2400(s.args, s.total_d, s.stream, s.events ? s.ev_scan0 : nullptr, s.events ? s.ev_scan1 : nullptr, gp));
    else HIP_TRY(c, launch_scan1090(s.args, s.total_d, s.stream, s.events ? s.ev_scan0 : nullptr, s.events ? s.ev_scan1 : nullptr, gp));
    if (attach) o.gather_launched = o.gather_attached = true;
    s.gather_launched = s.gather_attached = false;
    if (!s.args.total_chunks)
    { // nothing was launched: no records, no pass
        HIP_TRY(c, hipEventRecord(s.ev_order, s.stream));
        __atomic_store_n(s.total_h, (unsigned long long)(s.seq & 0x7FFFFFFFu) << 33, __ATOMIC_RELEASE);
        s.gather_launched = true;
    }
    return ADSB_AMD_OK;
}

// the slot's ordering pass as a launch of its own, unless a scan kernel has taken it on (enqueue) or it has been launched already
int ensure_gather(adsb_amd_ctx* c, Slot& s)
{
    if (s.gather_launched) return ADSB_AMD_OK;
    HIP_TRY(c, launch_gather1090(gather_args(s), s.stream, s.ev_order));
    s.gather_launched = true, s.gather_attached = false;
    return ADSB_AMD_OK;
}
} // namespace

extern "C" const char* adsb_amd_version(void) { return "libadsb_amd 0.1 (gfx950)"; }

// ---- control page of the node-shared record hand-over (shard.NodeGather): see adsb_amd.h
extern "C" void adsb_amd_shm_post_header(int64_t* slot, int64_t count, int64_t first_buffer, int64_t rank, int64_t step)
{
    __atomic_store_n(&slot[0], count, __ATOMIC_RELAXED);
    __atomic_store_n(&slot[1], first_buffer, __ATOMIC_RELAXED);
    __atomic_store_n(&slot[2], rank, __ATOMIC_RELAXED);
    __atomic_store_n(&slot[3], step, __ATOMIC_RELEASE); // last: whoever reads this step with acquire sees the three words and the records before them
}
extern "C" int adsb_amd_shm_read_header(const int64_t* slot, int64_t min_step, int64_t* out4)
{
    const int64_t step = __atomic_load_n(&slot[3], __ATOMIC_ACQUIRE);
    if (step < min_step) return 0;
    out4[0] = __atomic_load_n(&slot[0], __ATOMIC_RELAXED);
    out4[1] = __atomic_load_n(&slot[1], __ATOMIC_RELAXED);
    out4[2] = __atomic_load_n(&slot[2], __ATOMIC_RELAXED);
    out4[3] = step;
    return 1;
}
extern "C" void    adsb_amd_shm_store_release(int64_t* word, int64_t value) { __atomic_store_n(word, value, __ATOMIC_RELEASE); }
extern "C" int64_t adsb_amd_shm_load_acquire(const int64_t* word) { return __atomic_load_n(word, __ATOMIC_ACQUIRE); }

extern "C" int adsb_amd_create(adsb_amd_ctx_t** out, int device) { return adsb_amd_create_mode(out, device, ADSB_AMD_MODE_2000); }

extern "C" int adsb_amd_create_mode(adsb_amd_ctx_t** out, int device, int mode)
{
    if (!out) return ADSB_AMD_EINVAL;
    if (mode != ADSB_AMD_MODE_2000 && mode != ADSB_AMD_MODE_2400)
    {
        *out           = nullptr;
        g_create_error = "mode must be ADSB_AMD_MODE_2000 or ADSB_AMD_MODE_2400";
        return ADSB_AMD_EINVAL;
    }
    *out      = nullptr;
    int ndev  = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
    {
        g_create_error = std::string("no usable HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0")
                         + " (libadsb_amd has no CPU demodulation path)";
        return ADSB_AMD_ENODEV;
    }
    if (device < 0)
    {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= ndev)
    {
        g_create_error = "device index out of range";
        return ADSB_AMD_EINVAL;
    }
    adsb_amd_ctx* c = new (std::nothrow) adsb_amd_ctx();
    if (!c) return ADSB_AMD_EHIP;
    c->device = device;
    c->mode   = mode;
    auto bail = [&](const char* what, hipError_t err) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        adsb_amd_destroy(c);
        return ADSB_AMD_ENODEV;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    {
        // XCD-aware chunk ranges and the persistent grid follow the device's own topology (8 XCDs x 32 CUs on MI355X; other
        // partition modes expose fewer).  Only speed depends on these numbers, results do not.
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->ncu = (uint32_t)v;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, device) == hipSuccess && v > 0) c->nxcd = (uint32_t)v;
        else c->nxcd = 1;
        if (c->nxcd > kMaxXcd) c->nxcd = kMaxXcd;
    }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);

    uint32_t tab[112];
    build_crc_table(tab);
    if ((e = hipMalloc(&c->crc_tab, sizeof(tab))) != hipSuccess) return bail("hipMalloc(crc)", e);
    if ((e = hipMemcpy(c->crc_tab, tab, sizeof(tab), hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy(crc)", e);
    for (Slot& s : c->slot)
    {
        if ((e = hipMalloc(&s.total_d, kStateWords * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc(total)", e);
        if ((e = hipMemset(s.total_d, 0, kStateWords * sizeof(uint32_t))) != hipSuccess) return bail("hipMemset(total)", e);
        if ((e = hipMalloc(&s.work_d, kWorkCounters * kCounterStride * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc(work)", e);
        if ((e = hipMemset(s.work_d, 0, kWorkCounters * kCounterStride * sizeof(uint32_t))) != hipSuccess) return bail("hipMemset(work)", e);
        if ((e = hipHostMalloc(&s.total_h, sizeof(unsigned long long), hipHostMallocMapped)) != hipSuccess) return bail("hipHostMalloc(total)", e);
        *s.total_h = 0;
        if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&s.total_h_dev), s.total_h, 0)) != hipSuccess) return bail("hipHostGetDevicePointer(total)", e);
        // The three events ride on dispatches and order device work / give time stamps only; what the host reads (the count word, the
        // records) it reads after waiting for ev_order itself: no system-scope release when they are recorded (an event costs ~5 us of
        // stream time with it, ~3 without).
        if ((e = hipEventCreateWithFlags(&s.ev_scan0, hipEventDisableSystemFence)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&s.ev_scan1, hipEventDisableSystemFence)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&s.ev_order, hipEventDisableSystemFence)) != hipSuccess) return bail("hipEventCreate", e);
    }
    if constexpr (diag::kStamps)
    {
        if ((e = hipMalloc(&c->stamps_d, (size_t)kStampSteps * 4 * kStampGroups * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc(stamps)", e);
        if ((e = hipMemset(c->stamps_d, 0, (size_t)kStampSteps * 4 * kStampGroups * sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset(stamps)", e);
    }
    *out = c;
    return ADSB_AMD_OK;
}

#if DIAG_STAMPS
// measurement builds: every wave's clock values of one launch of the ring (4 x kStampGroups values: scan in, scan out, ordering pass in, out)
extern "C" int adsb_amd_debug_stamps_raw(adsb_amd_ctx_t* c, unsigned launch, unsigned long long* out)
{
    if (!c || !out) return ADSB_AMD_EINVAL;
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(out, c->stamps_d + (size_t)4 * kStampGroups * (launch % kStampSteps), (size_t)4 * kStampGroups * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return ADSB_AMD_OK;
}
// diagnostic builds: per launch of the ring {scan first in, scan last out, ordering pass first in, last out} (0 where nothing was noted), and
// how many launches there have been
extern "C" int adsb_amd_debug_stamps(adsb_amd_ctx_t* c, unsigned long long* out /* kStampSteps x 4 */, unsigned* launches)
{
    if (!c || !out || !launches) return ADSB_AMD_EINVAL;
    HIP_TRY(c, hipDeviceSynchronize());
    std::vector<unsigned long long> all((size_t)kStampSteps * 4 * kStampGroups);
    HIP_TRY(c, hipMemcpy(all.data(), c->stamps_d, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < (size_t)kStampSteps * 4; k++)
    {
        unsigned long long lo = ~0ull, hi = 0;
        for (size_t g = 0; g < kStampGroups; g++)
        {
            const unsigned long long v = all[k * kStampGroups + g];
            if (v == 0) continue;
            lo = std::min(lo, v);
            hi = std::max(hi, v);
        }
        out[k] = hi == 0 ? 0 : ((k & 1) ? hi : lo);
    }
    *launches = c->stamp_no;
    return ADSB_AMD_OK;
}
#endif

extern "C" void adsb_amd_destroy(adsb_amd_ctx_t* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (Slot& s : c->slot)
    {
        free_slot(s);
        if (s.total_d) (void)hipFree(s.total_d);
        if (s.work_d) (void)hipFree(s.work_d);
        if (s.total_h) (void)hipHostFree(s.total_h);
        if (s.host) (void)hipHostFree(s.host);
        if (s.host_dec) (void)hipHostFree(s.host_dec);
        if (s.host_packed) (void)hipHostFree(s.host_packed);
        if (s.ev_scan0) (void)hipEventDestroy(s.ev_scan0);
        if (s.ev_scan1) (void)hipEventDestroy(s.ev_scan1);
        if (s.ev_order) (void)hipEventDestroy(s.ev_order);
    }
    if (c->crc_tab) (void)hipFree(c->crc_tab);
    if (c->lut978) (void)hipFree(c->lut978);
    if (c->staging) (void)hipFree(c->staging);
    if (c->stamps_d) (void)hipFree(c->stamps_d);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c;
}

extern "C" const char* adsb_amd_last_error(const adsb_amd_ctx_t* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

extern "C" int adsb_amd_scan_1090_submit(adsb_amd_ctx_t* c, const void* iq_device, size_t nbytes, size_t buffer_bytes, void* hip_stream,
                                         int slot)
{
    if (!c) return ADSB_AMD_EINVAL;
    if (slot < 0 || slot >= kSlots) return fail(c, ADSB_AMD_EINVAL, "slot must be 0, 1 or 2");
    Slot& s = c->slot[slot];
    if (s.pending) return fail(c, ADSB_AMD_ESTATE, "slot still has an unfetched scan");
    HIP_TRY(c, hipSetDevice(c->device));
    ScanArgs a;
    int      rc = make_args(c, iq_device, nbytes, buffer_bytes, &a);
    if (rc) return rc;
    size_t cap = s.cap_per_chunk ? s.cap_per_chunk : kDefaultCap;
    if (s.cap_hint > cap) cap = s.cap_hint; // another slot had to grow its regions for this kind of input
    if ((rc = ensure_slot(c, s, a.total_chunks, cap, c->outputs))) return rc;
    s.produced = c->outputs;
    s.args     = a;
    s.stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    if ((rc = enqueue(c, s))) return rc;
    s.pending = true;
    return ADSB_AMD_OK;
}

namespace
{
// Wait for the slot's scan; when a chunk produced more records than its region holds (dense noise, adversarial input) repeat the scan
// with regions eight times larger, up to the hard bound of two records per preamble position -- as long as the record arrays
// (regions + dense + decoded) still fit in free device memory; beyond that the call fails with ADSB_AMD_ENOMEM instead of leaning on
// hipMalloc to refuse.  The region size that worked is remembered for EVERY slot (the others start their next scans with it).
// The slot's record count and overflow flag -> s.nrecords, s.overflow.
int wait_count(adsb_amd_ctx* c, Slot& s)
{
    // Poll the stamp.  The word is written once per launch, by one store; the pass's event is asked now and then so that a launch that
    // failed (the event completes in error, or completes without the word ever arriving) ends the wait instead of hanging it.
    {
        const int rc = ensure_gather(c, s);
        if (rc) return rc;
    }
    const unsigned long long want = s.seq & 0x7FFFFFFFu;
    for (unsigned spins = 0;; spins++)
    {
        const unsigned long long w = __atomic_load_n(s.total_h, __ATOMIC_ACQUIRE);
        if ((w >> 33) == want)
        {
            s.nrecords = (uint32_t)w;
            s.overflow = ((w >> 32) & 1u) != 0;
            return ADSB_AMD_OK;
        }
        if ((spins & 0x3FFu) == 0x3FFu)
        {
            // (a pass that rides in the other slot's scan kernel has no event of its own: the stream it is on is asked instead)
            const hipError_t q = s.gather_attached ? hipStreamQuery(s.stream) : hipEventQuery(s.ev_order);
            if (q == hipSuccess)
            { // the pass is done: its store has left the device; give it a moment to land, then it is an error
                for (int k = 0; k < 1000000; k++)
                {
                    const unsigned long long w2 = __atomic_load_n(s.total_h, __ATOMIC_ACQUIRE);
                    if ((w2 >> 33) == want)
                    {
                        s.nrecords = (uint32_t)w2;
                        s.overflow = ((w2 >> 32) & 1u) != 0;
                        return ADSB_AMD_OK;
                    }
                    __builtin_ia32_pause();
                }
                return fail(c, ADSB_AMD_EHIP, "the ordering pass finished without delivering its record count");
            }
            if (q != hipErrorNotReady)
            {
                c->error = std::string("scan failed: ") + hipGetErrorString(q);
                return ADSB_AMD_EHIP;
            }
        }
        __builtin_ia32_pause();
    }
}

int wait_scan(adsb_amd_ctx* c, Slot& s)
{
    for (;;)
    {
        {
            const int rc = wait_count(c, s);
            if (rc) return rc;
        }
        if (!s.overflow) break;
        // (the count comes from the pass's last finisher: the other sum array and the work counters have been zeroed, nothing of the pass is still running)
        if (!s.gather_attached) HIP_TRY(c, hipEventSynchronize(s.ev_order));
        size_t cap = s.cap_per_chunk * 8;
        if (cap > (size_t)2 * kChunk) cap = (size_t)2 * kChunk;
        if (cap == s.cap_per_chunk) return fail(c, ADSB_AMD_EHIP, "record overflow at the maximum region size (internal error)");
        const size_t nch  = s.args.total_chunks ? s.args.total_chunks : 1;
        const size_t per  = 2 * sizeof(adsb_amd_record_t) + sizeof(adsb_amd_decoded_t) + ((s.produced & ADSB_AMD_OUT_PACKED) ? sizeof(adsb_amd_packed_t) : 0); // regions + dense + decoded (+ packed)
        const size_t need = nch * cap * per, have = s.chunks_cap * s.cap_per_chunk * per;
        size_t       free_b = 0, total_b = 0;
        HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
        if (need > free_b + have)
        {
            c->error = "record regions would need " + std::to_string(need >> 20) + " MiB (" + std::to_string(cap) + " records per chunk), device has " +
                       std::to_string((free_b + have) >> 20) + " MiB free: split the input into smaller scan calls";
            return ADSB_AMD_ENOMEM;
        }
        int rc = ensure_slot(c, s, s.args.total_chunks, cap, s.produced);
        if (rc == ADSB_AMD_OK) rc = enqueue(c, s);
        if (rc) return rc;
        for (Slot& other : c->slot)
            if (&other != &s && other.cap_per_chunk < cap && other.cap_hint < cap) other.cap_hint = cap;
    }
    return ADSB_AMD_OK;
}

// The scan kernel's and the whole submit's device time of a slot whose pass has been waited for (the scan is the first thing a submit
// enqueues).  A scan without the two events, or events the runtime cannot read, leave the slot untimed.
void read_timing(Slot& s)
{
    s.timed = s.events && hipEventElapsedTime(&s.scan_ms, s.ev_scan0, s.ev_scan1) == hipSuccess;
    if (s.timed && s.gather_attached) s.total_ms = s.scan_ms; // (the pass was part of a later kernel: no event marks its end)
    else s.timed = s.timed && hipEventElapsedTime(&s.total_ms, s.ev_scan0, s.ev_order) == hipSuccess;
}

// Body of fetch; the caller clears `pending` whatever the outcome, so a failed sync or copy never wedges the slot.  `what`: ADSB_AMD_OUT_* to
// bring to the host.
int fetch_slot(adsb_amd_ctx* c, Slot& s, unsigned what)
{
    HIP_TRY(c, hipSetDevice(c->device));
    if (what & ~s.produced) return fail(c, ADSB_AMD_ESTATE, "this slot's scan did not produce the array asked for (adsb_amd_set_outputs before the submit)");
    {
        const int rc = wait_scan(c, s);
        if (rc) return rc;
    }
    int rc = ensure_host(c, s, s.nrecords);
    if (rc) return rc;
    // Round 6: the count is stored by the pass's LAST FINISHER -- an atomic ticket over its blocks -- after every wave of the pass has had its
    // stores acknowledged, and the records are written through to memory (gather1090.hip.h): seeing the stamp means the records can be copied and the
    // slot resubmitted.  (Round 5 stored it from the last workgroup by index and had to wait for the pass's event on top; a stream-side wait on that
    // re-used, dispatch-riding event had let stale records through.)  A pass launched on its own still has its event waited for: it is there, and
    // the slot's timing reads it.
    if (!s.gather_attached) HIP_TRY(c, hipEventSynchronize(s.ev_order));
    if (s.nrecords)
    {
        if (what & ADSB_AMD_OUT_RECORDS)
            HIP_TRY(c, hipMemcpyAsync(s.host, s.dense, s.nrecords * sizeof(adsb_amd_record_t), hipMemcpyDeviceToHost, c->copy_stream));
        if (what & ADSB_AMD_OUT_DECODED)
            HIP_TRY(c, hipMemcpyAsync(s.host_dec, s.decoded, s.nrecords * sizeof(adsb_amd_decoded_t), hipMemcpyDeviceToHost, c->copy_stream));
        if (what & ADSB_AMD_OUT_PACKED)
            HIP_TRY(c, hipMemcpyAsync(s.host_packed, s.packed, s.nrecords * sizeof(adsb_amd_packed_t), hipMemcpyDeviceToHost, c->copy_stream));
        HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    }
    s.dec_valid = (what & ADSB_AMD_OUT_DECODED) != 0;
    read_timing(s);
    return ADSB_AMD_OK;
}
} // namespace

extern "C" int adsb_amd_set_outputs(adsb_amd_ctx_t* c, unsigned mask)
{
    if (!c) return ADSB_AMD_EINVAL;
    if (mask == 0 || (mask & ~(ADSB_AMD_OUT_RECORDS | ADSB_AMD_OUT_DECODED | ADSB_AMD_OUT_PACKED))) return fail(c, ADSB_AMD_EINVAL, "outputs: a non-empty combination of ADSB_AMD_OUT_*");
    c->outputs = mask;
    return ADSB_AMD_OK;
}

extern "C" int adsb_amd_set_timing(adsb_amd_ctx_t* c, unsigned every)
{
    if (!c) return ADSB_AMD_EINVAL;
    c->timing_every = every;
    c->submits      = 0;
    return ADSB_AMD_OK;
}

extern "C" int adsb_amd_scan_1090_fetch_packed(adsb_amd_ctx_t* c, int slot, const adsb_amd_packed_t** packed, size_t* n)
{
    if (!c) return ADSB_AMD_EINVAL;
    if (slot < 0 || slot >= kSlots) return fail(c, ADSB_AMD_EINVAL, "slot must be 0, 1 or 2");
    Slot& s = c->slot[slot];
    if (!s.pending) return fail(c, ADSB_AMD_ESTATE, "fetch without submit");
    const int rc = fetch_slot(c, s, ADSB_AMD_OUT_PACKED);
    s.pending    = false;
    if (rc) return rc;
    if (packed) *packed = s.host_packed;
    if (n) *n = s.nrecords;
    return ADSB_AMD_OK;
}

/* The packed fetch in two halves (round 6).  _begin waits for the slot's scan and its ordering pass, puts the copy of the packed records on the copy
 * stream and returns with the slot free for its next submit: a scan only writes the slot's raw record regions, the dense arrays the copy reads are
 * written by the ordering pass of THAT scan, which runs in front of the scan kernel after it or in adsb_amd_scan_1090_fetch* -- after _end in a loop
 * that ends what it begins.  _end waits for the copy.  Between the two the caller submits the slot's next scan: since the ordering pass of a scan
 * rides in front of the NEXT scan kernel on its stream, its count reaches the host some tens of microseconds into that kernel, and a loop that had to
 * wait for the 0.16 ms copy before it could submit again would leave the GPU waiting for it. */
extern "C" int adsb_amd_scan_1090_fetch_packed_begin(adsb_amd_ctx_t* c, int slot, size_t* n)
{
    if (!c) return ADSB_AMD_EINVAL;
    if (slot < 0 || slot >= kSlots) return fail(c, ADSB_AMD_EINVAL, "slot must be 0, 1 or 2");
    Slot& s = c->slot[slot];
    if (!s.pending) return fail(c, ADSB_AMD_ESTATE, "fetch without submit");
    if (s.copying) return fail(c, ADSB_AMD_ESTATE, "the slot's last fetch has been begun and not ended");
    auto body = [&]() -> int {
        HIP_TRY(c, hipSetDevice(c->device));
        if (!(s.produced & ADSB_AMD_OUT_PACKED)) return fail(c, ADSB_AMD_ESTATE, "this slot's scan did not produce the packed form (adsb_amd_set_outputs before the submit)");
        int rc = wait_scan(c, s);
        if (rc) return rc;
        if ((rc = ensure_host(c, s, s.nrecords))) return rc;
        if (!s.gather_attached) HIP_TRY(c, hipEventSynchronize(s.ev_order)); // (see fetch_slot)
        if (s.nrecords) HIP_TRY(c, hipMemcpyAsync(s.host_packed, s.packed, s.nrecords * sizeof(adsb_amd_packed_t), hipMemcpyDeviceToHost, c->copy_stream));
        read_timing(s);
        s.copy_n = s.nrecords, s.copying = true;
        return ADSB_AMD_OK;
    };
    const int rc = body();
    s.pending    = false;
    if (rc == ADSB_AMD_OK && n) *n = s.copy_n;
    return rc;
}
extern "C" int adsb_amd_scan_1090_fetch_packed_end(adsb_amd_ctx_t* c, int slot, const adsb_amd_packed_t** packed, size_t* n)
{
    if (!c) return ADSB_AMD_EINVAL;
    if (slot < 0 || slot >= kSlots) return fail(c, ADSB_AMD_EINVAL, "slot must be 0, 1 or 2");
    Slot& s = c->slot[slot];
    if (!s.copying) return fail(c, ADSB_AMD_ESTATE, "no fetch begun on this slot");
    s.copying = false;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    if (packed) *packed = s.host_packed;
    if (n) *n = s.copy_n;
    return ADSB_AMD_OK;
}

/* The returned pointer aims into the slot's page-locked result buffer: it stays valid until the next submit on this slot
 * (a later fetch may have to grow that buffer, which moves it). */
extern "C" int adsb_amd_scan_1090_fetch(adsb_amd_ctx_t* c, int slot, const adsb_amd_record_t** records, size_t* n)
{
    if (!c) return ADSB_AMD_EINVAL;
    if (slot < 0 || slot >= kSlots) return fail(c, ADSB_AMD_EINVAL, "slot must be 0, 1 or 2");
    Slot& s = c->slot[slot];
    if (!s.pending) return fail(c, ADSB_AMD_ESTATE, "fetch without submit");
    const int rc = fetch_slot(c, s, ADSB_AMD_OUT_RECORDS);
    s.pending    = false;
    if (rc) return rc;
    if (records) *records = s.host;
    if (n) *n = s.nrecords;
    return ADSB_AMD_OK;
}

extern "C" int adsb_amd_scan_1090_fetch_decoded(adsb_amd_ctx_t* c, int slot, const adsb_amd_record_t** records, const adsb_amd_decoded_t** decoded,
                                                size_t* n)
{
    if (!c) return ADSB_AMD_EINVAL;
    if (slot < 0 || slot >= kSlots) return fail(c, ADSB_AMD_EINVAL, "slot must be 0, 1 or 2");
    Slot& s = c->slot[slot];
    if (!s.pending) return fail(c, ADSB_AMD_ESTATE, "fetch without submit");
    const int rc = fetch_slot(c, s, ADSB_AMD_OUT_RECORDS | ADSB_AMD_OUT_DECODED);
    s.pending    = false;
    if (rc) return rc;
    if (records) *records = s.host;
    if (decoded) *decoded = s.host_dec;
    if (n) *n = s.nrecords;
    return ADSB_AMD_OK;
}

/* Device-side delivery of a slot's result: waits for the scan and copies its sorted records into `dst_device` (device memory
 * of the same GPU, room for `cap` records) on `hip_stream` (NULL: the context's copy stream; the call returns after the copy
 * has been enqueued there, and synchronises only in the NULL case).  For consumers that keep working on the GPU -- the
 * multi-GPU gather sends the records to the root rank over RCCL without a host bounce.  The slot stays fetched afterwards. */
namespace
{
int fetch_device_impl(adsb_amd_ctx_t* c, int slot, void* dst_device, size_t cap, void* hip_stream, size_t* n, bool packed);
}
extern "C" int adsb_amd_scan_1090_fetch_device(adsb_amd_ctx_t* c, int slot, void* dst_device, size_t cap, void* hip_stream, size_t* n)
{
    return fetch_device_impl(c, slot, dst_device, cap, hip_stream, n, false);
}
extern "C" int adsb_amd_scan_1090_fetch_device_packed(adsb_amd_ctx_t* c, int slot, void* dst_device, size_t cap, void* hip_stream, size_t* n)
{
    return fetch_device_impl(c, slot, dst_device, cap, hip_stream, n, true);
}
namespace
{
int fetch_device_impl(adsb_amd_ctx_t* c, int slot, void* dst_device, size_t cap, void* hip_stream, size_t* n, bool packed)
{
    if (!c) return ADSB_AMD_EINVAL;
    if (slot < 0 || slot >= kSlots) return fail(c, ADSB_AMD_EINVAL, "slot must be 0, 1 or 2");
    Slot& s = c->slot[slot];
    if (!s.pending) return fail(c, ADSB_AMD_ESTATE, "fetch without submit");
    auto body = [&]() -> int {
        HIP_TRY(c, hipSetDevice(c->device));
        {
            const int rc = wait_scan(c, s); // repeats the scan with larger regions when a chunk overflowed its region
            if (rc) return rc;
        }
        if (n) *n = s.nrecords;
        if (!(s.produced & (packed ? ADSB_AMD_OUT_PACKED : ADSB_AMD_OUT_RECORDS)))
            return fail(c, ADSB_AMD_ESTATE, "this slot's scan did not produce the array asked for (adsb_amd_set_outputs)");
        if (s.nrecords > cap) return fail(c, ADSB_AMD_ENOSPC, "destination too small");
        if (!s.gather_attached) HIP_TRY(c, hipEventSynchronize(s.ev_order)); // (see fetch_slot)
        if (s.nrecords)
        {
            hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->copy_stream;
            // hipMemcpyDefault: the destination may be device memory or page-locked / registered host memory (shard.NodeGather)
            static_assert(sizeof(adsb_amd_record_t) == sizeof(adsb_amd_packed_t), "one size for both forms");
            HIP_TRY(c, hipMemcpyAsync(dst_device, packed ? static_cast<const void*>(s.packed) : static_cast<const void*>(s.dense),
                                      s.nrecords * sizeof(adsb_amd_record_t), hipMemcpyDefault, st));
            if (!hip_stream) HIP_TRY(c, hipStreamSynchronize(st));
        }
        read_timing(s);
        return ADSB_AMD_OK;
    };
    const int rc = body();
    s.pending    = false;
    return rc;
}
} // namespace

extern "C" int adsb_amd_scan_1090_timing(adsb_amd_ctx_t* c, int slot, float* scan_kernel_ms, float* total_ms)
{
    if (!c) return ADSB_AMD_EINVAL;
    if (slot < 0 || slot >= kSlots) return fail(c, ADSB_AMD_EINVAL, "slot must be 0, 1 or 2");
    Slot& s = c->slot[slot];
    if (!s.timed) return fail(c, ADSB_AMD_ESTATE, "no completed, timed scan on this slot (adsb_amd_set_timing)");
    if (scan_kernel_ms) *scan_kernel_ms = s.scan_ms;
    if (total_ms) *total_ms = s.total_ms;
    return ADSB_AMD_OK;
}

namespace
{
int stage_input(adsb_amd_ctx* c, const uint8_t* iq_host, size_t nbytes)
{
    size_t want = nbytes + 64;
    if (want > c->staging_cap)
    {
        if (c->staging) (void)hipFree(c->staging);
        c->staging     = nullptr;
        c->staging_cap = 0;
        HIP_TRY(c, hipMalloc(&c->staging, want));
        c->staging_cap = want;
    }
    if (nbytes) HIP_TRY(c, hipMemcpyAsync(c->staging, iq_host, nbytes, hipMemcpyHostToDevice, c->stream));
    return ADSB_AMD_OK;
}
} // namespace

extern "C" int adsb_amd_scan_1090(adsb_amd_ctx_t* c, const uint8_t* iq_host, size_t nbytes, size_t buffer_bytes, adsb_amd_record_t* out,
                                  size_t cap, size_t* n_out)
{
    if (!c) return ADSB_AMD_EINVAL;
    if (!iq_host && nbytes) return fail(c, ADSB_AMD_EINVAL, "iq_host is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = stage_input(c, iq_host, nbytes);
    if (rc) return rc;
    if ((rc = adsb_amd_scan_1090_submit(c, c->staging, nbytes, buffer_bytes, c->stream, 0))) return rc;
    const adsb_amd_record_t* rec = nullptr;
    size_t                   n   = 0;
    if ((rc = adsb_amd_scan_1090_fetch(c, 0, &rec, &n))) return rc;
    if (n_out) *n_out = n;
    if (n > cap) return fail(c, ADSB_AMD_ENOSPC, "output array too small");
    if (n && out) std::memcpy(out, rec, n * sizeof(adsb_amd_record_t));
    return ADSB_AMD_OK;
}

extern "C" int adsb_amd_magnitude_1090(adsb_amd_ctx_t* c, const uint8_t* iq_host, size_t nbytes, uint16_t* mag_out)
{
    if (!c) return ADSB_AMD_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    size_t n = nbytes / 2;
    if (n == 0) return ADSB_AMD_OK;
    int rc = stage_input(c, iq_host, nbytes);
    if (rc) return rc;
    uint16_t* d = nullptr;
    HIP_TRY(c, hipMalloc(&d, n * sizeof(uint16_t)));
    hipError_t e = launch_magnitude1090(c->staging, d, n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(mag_out, d, n * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess)
    {
        c->error = std::string("magnitude_1090: ") + hipGetErrorString(e);
        return ADSB_AMD_EHIP;
    }
    return ADSB_AMD_OK;
}

// ------------------------------------------------------------------------------------------------ handler
namespace
{
// the handler's entry points choose the ordering pass's outputs for their own scans and put the caller's choice back when they return
struct OutputsGuard
{
    adsb_amd_ctx* c;
    unsigned      saved;
    OutputsGuard(adsb_amd_ctx* ctx, unsigned mask) : c(ctx), saved(ctx->outputs) { c->outputs = mask; }
    ~OutputsGuard() { c->outputs = saved; }
};
} // namespace

// Staging of the recorded-file replay (adsb_amd_handler_replay_file): the handler's, set up by the first replay and kept -- libadsb replays a
// file over and over (RTLSDR.hpp:419-442 re-opens it until stopped), and page-locking and releasing the buffers is a fifth of one pass.
struct ReplayStaging
{
    static constexpr int    kPinned = 4, kStage = 3;
    static constexpr size_t kSlice  = 64; // buffers per batch (16 MiB)
    uint8_t*                stage[kStage]   = {};
    uint8_t*                pinned[kPinned] = {};
    hipStream_t             up_stream       = nullptr;
    bool reserve(size_t nbatch)
    {
        const size_t bytes = kSlice * ADSB_AMD_REF_BUFFER_BYTES;
        bool         ok    = up_stream || hipStreamCreateWithFlags(&up_stream, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; ok && i < kStage && (size_t)i < nbatch; i++) ok = stage[i] || hipMalloc(&stage[i], bytes) == hipSuccess;
        for (int i = 0; ok && i < kPinned && (size_t)i < nbatch; i++) ok = pinned[i] || hipHostMalloc(&pinned[i], bytes, hipHostMallocDefault) == hipSuccess;
        return ok;
    }
    void release()
    {
        for (uint8_t*& p : stage)
            if (p) (void)hipFree(p), p = nullptr;
        for (uint8_t*& p : pinned)
            if (p) (void)hipHostFree(p), p = nullptr;
        if (up_stream) (void)hipStreamDestroy(up_stream), up_stream = nullptr;
    }
};

struct adsb_amd_handler
{
    adsb_amd_ctx*                  ctx = nullptr;
    adsb_amd::Resolver1090         resolver;
    std::vector<adsb_amd_record_t> scratch;
    bool                           want_frames = true; // false: records travel in the packed form, callbacks get frames without message bytes
    std::string                    error;
    ReplayStaging                  replay;
};

extern "C" int adsb_amd_handler_create(adsb_amd_handler_t** out, int device) { return adsb_amd_handler_create_mode(out, device, ADSB_AMD_MODE_2000); }

extern "C" int adsb_amd_handler_create_mode(adsb_amd_handler_t** out, int device, int mode)
{
    if (!out) return ADSB_AMD_EINVAL;
    *out                = nullptr;
    adsb_amd_handler* h = new (std::nothrow) adsb_amd_handler();
    if (!h) return ADSB_AMD_EHIP;
    h->resolver.set_mode(mode);
    int rc = adsb_amd_create_mode(&h->ctx, device, mode);
    if (rc)
    {
        delete h;
        return rc;
    }
    *out = h;
    return ADSB_AMD_OK;
}

extern "C" void adsb_amd_handler_destroy(adsb_amd_handler_t* h)
{
    if (!h) return;
    if (h->ctx && hipSetDevice(h->ctx->device) == hipSuccess) h->replay.release();
    adsb_amd_destroy(h->ctx);
    delete h;
}

extern "C" const char* adsb_amd_handler_last_error(const adsb_amd_handler_t* h)
{
    if (!h) return adsb_amd_last_error(nullptr);
    return h->error.empty() ? adsb_amd_last_error(h->ctx) : h->error.c_str();
}

extern "C" void adsb_amd_handler_set_sample_clock(adsb_amd_handler_t* h, int64_t t0_ns, uint32_t rate_hz)
{
    if (h) h->resolver.set_sample_clock(t0_ns, rate_hz);
}

extern "C" long adsb_amd_handler_handle_data(adsb_amd_handler_t* h, const uint8_t* iq_host, size_t nbytes, size_t buffer_bytes,
                                             adsb_amd_on_changed_fn cb, void* user)
{
    if (!h) return ADSB_AMD_EINVAL;
    h->error.clear();
    adsb_amd_ctx* c = h->ctx;
    if (!iq_host && nbytes) return fail(c, ADSB_AMD_EINVAL, "iq_host is NULL");
    // libadsb calls HandleData from its transport's consumer thread (RTLSDR.hpp:470-473) and HIP's current device is per
    // thread: select this handler's device before anything is allocated or copied
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = stage_input(c, iq_host, nbytes);
    if (rc) return rc;
    const OutputsGuard guard(c, h->want_frames ? (ADSB_AMD_OUT_RECORDS | ADSB_AMD_OUT_DECODED) : ADSB_AMD_OUT_PACKED);
    if ((rc = adsb_amd_scan_1090_submit(c, c->staging, nbytes, buffer_bytes, c->stream, 0))) return rc;
    const adsb_amd_record_t*  rec = nullptr;
    const adsb_amd_decoded_t* dec = nullptr;
    const adsb_amd_packed_t*  pk  = nullptr;
    size_t                    n   = 0;
    if (h->want_frames) rc = adsb_amd_scan_1090_fetch_decoded(c, 0, &rec, &dec, &n);
    else rc = adsb_amd_scan_1090_fetch_packed(c, 0, &pk, &n);
    if (rc) return rc;
    const ScanArgs& a = c->slot[0].args;
    // stream position advances by everything the caller handed over, as the reference's HandleData consumes it
    size_t spb  = a.nbuf ? a.buf_samples : nbytes / 2;
    size_t nbuf = a.nbuf ? a.nbuf : 1;
    return h->want_frames ? h->resolver.feed(rec, dec, n, spb, nbuf, cb, user) : h->resolver.feed_packed(pk, n, spb, nbuf, cb, user);
}

extern "C" void adsb_amd_handler_set_frames(adsb_amd_handler_t* h, int want_frames)
{
    if (h) h->want_frames = want_frames != 0;
}

// Recorded-file replay in batches: what RTLSDR::TestDataReadLoop (RTLSDR.hpp:419-442) feeds a handler -- whole BufferLength
// (262 144 B) reads in file order, each buffer demodulated on its own, a trailing partial read never delivered -- for one
// pass over buffers [first_buffer, first_buffer + max_buffers) of the file (the reference re-opens the file and loops until
// stopped; ranks of a multi-GPU job take disjoint ranges).  The file is mapped; batches of 256 buffers go through a two-stage
// pipeline (upload of batch k+1 beside scan of k and resolve of k-1).  adsb_amd_handler_run_replay below is the same file through
// the ring, one buffer per HandleData, the way libadsb's replay mode paces it.
extern "C" long adsb_amd_handler_replay_file(adsb_amd_handler_t* h, const char* path, size_t first_buffer, size_t max_buffers,
                                             adsb_amd_on_changed_fn cb, void* user)
{
    if (!h || !path) return ADSB_AMD_EINVAL;
    h->error.clear();
    const int fd = open(path, O_RDONLY);
    if (fd < 0)
    {
        h->error = std::string("cannot open ") + path + ": " + strerror(errno);
        return ADSB_AMD_EINVAL;
    }
    struct stat st;
    if (fstat(fd, &st) != 0)
    {
        h->error = std::string("fstat ") + path + ": " + strerror(errno);
        close(fd);
        return ADSB_AMD_EINVAL;
    }
    const size_t BB    = ADSB_AMD_REF_BUFFER_BYTES;
    const size_t total = (size_t)st.st_size / BB; // whole buffers only
    if (first_buffer >= total || max_buffers == 0)
    {
        close(fd);
        return 0;
    }
    const size_t nbuf = std::min(max_buffers, total - first_buffer);
    constexpr size_t kSlice = ReplayStaging::kSlice;
    const size_t     nbatch = (nbuf + kSlice - 1) / kSlice;
    adsb_amd_ctx*    c      = h->ctx;
    // Three stages beside this thread, each a batch ahead of the next: reader threads copy batch k+2 out of the page cache into page-locked
    // memory (pread straight into it: a copy at memory speed; handing the runtime a mapping of the file instead was measured at 7.7 GB/s),
    // the uploader sends batch k+1 to the device (one DMA), the GPU scans batch k and this thread resolves batch k-1.  Until round 6 the
    // reading and the upload of a batch ran one after the other on one thread, eight reader threads were started per batch, the batches
    // were 64 MiB (256 MiB of page-locked and device memory set up and released per call), and with two device buffers an upload could
    // not start before the batch two behind it was resolved: 68 ms per GiB, now 29-31 (profiles/r06_replay.txt).  Buffers stay independent
    // (a batch is a whole number of them), the resolver sees the batches in file order, so the callback stream is the one of
    // buffer-by-buffer delivery.
    constexpr int   kPinned = ReplayStaging::kPinned, kStage = ReplayStaging::kStage, kReaders = 8;
    long            accepted = 0;
    if (hipSetDevice(c->device) != hipSuccess || !h->replay.reserve(nbatch))
    {
        h->error = "replay_file: cannot allocate the staging buffers";
        h->replay.release();
        close(fd);
        return ADSB_AMD_EHIP;
    }
    uint8_t* const* const stage     = h->replay.stage;
    uint8_t* const* const pinned    = h->replay.pinned;
    const hipStream_t     up_stream = h->replay.up_stream;
    std::mutex              mu;
    std::condition_variable cv;
    std::vector<int>        read_parts(nbatch, 0);      // reader threads done with their part of a batch
    size_t                  uploaded = 0, released = 0; // batches whose upload is complete / whose device staging buffer is free again
    std::atomic<bool>       failed{false};
    auto                    fail = [&]() {
        std::unique_lock lk(mu);
        failed = true;
        cv.notify_all();
    };
    std::thread readers[kReaders];
    for (int t = 0; t < kReaders; t++)
        readers[t] = std::thread([&, t]() {
            for (size_t b = 0; b < nbatch && !failed; b++)
            {
                {
                    std::unique_lock lk(mu);
                    cv.wait(lk, [&]() { return b < uploaded + (size_t)kPinned || failed; }); // the page-locked buffer's last upload is done
                    if (failed) return;
                }
                const size_t bytes = std::min(kSlice, nbuf - b * kSlice) * BB;
                const off_t  at    = (off_t)((first_buffer + b * kSlice) * BB);
                size_t       lo = bytes * (size_t)t / kReaders, hi = bytes * (size_t)(t + 1) / kReaders;
                while (lo < hi)
                {
                    const ssize_t r = pread(fd, pinned[b % kPinned] + lo, hi - lo, at + (off_t)lo);
                    if (r <= 0) return fail();
                    lo += (size_t)r;
                }
                std::unique_lock lk(mu);
                if (++read_parts[b] == kReaders) cv.notify_all();
            }
        });
    std::thread uploader([&]() {
        if (hipSetDevice(c->device) != hipSuccess) return fail();
        for (size_t b = 0; b < nbatch && !failed; b++)
        {
            {
                std::unique_lock lk(mu);
                cv.wait(lk, [&]() { return (read_parts[b] == kReaders && b < released + (size_t)kStage) || failed; });
                if (failed) return;
            }
            const size_t bytes = std::min(kSlice, nbuf - b * kSlice) * BB;
            if (hipMemcpyAsync(stage[b % kStage], pinned[b % kPinned], bytes, hipMemcpyHostToDevice, up_stream) != hipSuccess ||
                hipStreamSynchronize(up_stream) != hipSuccess)
                return fail();
            std::unique_lock lk(mu);
            uploaded = b + 1;
            cv.notify_all();
        }
    });
    auto resolve = [&](size_t b) -> int { // batch b was submitted on slot b & 1
        const adsb_amd_record_t*  rec = nullptr;
        const adsb_amd_decoded_t* dec = nullptr;
        const adsb_amd_packed_t*  pk  = nullptr;
        size_t                    nr  = 0;
        int rc = h->want_frames ? adsb_amd_scan_1090_fetch_decoded(c, (int)(b & 1), &rec, &dec, &nr) : adsb_amd_scan_1090_fetch_packed(c, (int)(b & 1), &pk, &nr);
        if (rc) return rc;
        const size_t n = std::min(kSlice, nbuf - b * kSlice);
        h->resolver.set_frame_offset_base(b * kSlice * (BB / 2)); // a frame's offset counts from the first buffer of the pass, whatever the batches are
        const long   a = h->want_frames ? h->resolver.feed(rec, dec, nr, BB / 2, n, cb, user) : h->resolver.feed_packed(pk, nr, BB / 2, n, cb, user);
        h->resolver.set_frame_offset_base(0);
        if (a < 0) return (int)a;
        accepted += a;
        std::unique_lock lk(mu);
        released = b + 1;
        cv.notify_all();
        return 0;
    };
    int rc = 0;
    const OutputsGuard guard(c, h->want_frames ? (ADSB_AMD_OUT_RECORDS | ADSB_AMD_OUT_DECODED) : ADSB_AMD_OUT_PACKED);
    for (size_t b = 0; b < nbatch && !rc; b++)
    {
        {
            std::unique_lock lk(mu);
            cv.wait(lk, [&]() { return uploaded > b || failed; });
            if (failed) break;
        }
        const size_t n = std::min(kSlice, nbuf - b * kSlice);
        rc             = adsb_amd_scan_1090_submit(c, stage[b % kStage], n * BB, BB, c->stream, (int)(b & 1));
        if (!rc && b > 0) rc = resolve(b - 1);
    }
    if (!rc && !failed && nbatch > 0) rc = resolve(nbatch - 1);
    {
        std::unique_lock lk(mu);
        if (rc) failed = true;
        cv.notify_all();
    }
    uploader.join();
    for (auto& t : readers) t.join();
    (void)hipDeviceSynchronize();
    for (Slot& sl : c->slot) sl.pending = false; // a failed run may leave a submitted scan behind
    if (failed && !rc)
    {
        h->error = "replay_file: upload failed";
        rc       = ADSB_AMD_EHIP;
    }
    close(fd);
    return rc ? rc : accepted;
}

// Transport helper: a host that keeps its IQ ring (RTLSDR.hpp:564-570 keeps BufferCount slots of BufferLength bytes) in
// page-locked memory lets every HandleData upload by DMA straight out of the slot instead of through the runtime's pageable
// staging path.  The host does not have to link HIP for that.
extern "C" int adsb_amd_host_alloc(void** out, size_t nbytes)
{
    if (!out || nbytes == 0) return ADSB_AMD_EINVAL;
    *out = nullptr;
    return hipHostMalloc(out, nbytes, hipHostMallocDefault) == hipSuccess ? ADSB_AMD_OK : ADSB_AMD_EHIP;
}
extern "C" void adsb_amd_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

// ------------------------------------------------------------------------------------------------ resolver (host only)
struct adsb_amd_resolver
{
    adsb_amd::Resolver1090 impl;
};

extern "C" adsb_amd_resolver_t* adsb_amd_resolver_create(void) { return new (std::nothrow) adsb_amd_resolver(); }
extern "C" void                 adsb_amd_resolver_destroy(adsb_amd_resolver_t* r) { delete r; }
extern "C" void adsb_amd_resolver_set_sample_clock(adsb_amd_resolver_t* r, int64_t t0_ns, uint32_t rate_hz)
{
    if (r) r->impl.set_sample_clock(t0_ns, rate_hz);
}
extern "C" void adsb_amd_resolver_set_mode(adsb_amd_resolver_t* r, int mode)
{
    if (r) r->impl.set_mode(mode);
}
extern "C" long adsb_amd_resolver_feed(adsb_amd_resolver_t* r, const adsb_amd_record_t* records, size_t n, size_t samples_per_buffer,
                                       size_t nbuffers, adsb_amd_on_changed_fn cb, void* user)
{
    if (!r) return ADSB_AMD_EINVAL;
    return r->impl.feed(records, nullptr, n, samples_per_buffer, nbuffers, cb, user);
}
extern "C" long adsb_amd_resolver_feed_decoded(adsb_amd_resolver_t* r, const adsb_amd_record_t* records, const adsb_amd_decoded_t* decoded, size_t n,
                                               size_t samples_per_buffer, size_t nbuffers, adsb_amd_on_changed_fn cb, void* user)
{
    if (!r || (n && !decoded)) return ADSB_AMD_EINVAL;
    return r->impl.feed(records, decoded, n, samples_per_buffer, nbuffers, cb, user);
}
extern "C" long adsb_amd_resolver_feed_packed(adsb_amd_resolver_t* r, const adsb_amd_packed_t* packed, size_t n, size_t samples_per_buffer, size_t nbuffers,
                                              adsb_amd_on_changed_fn cb, void* user)
{
    if (!r || (n && !packed)) return ADSB_AMD_EINVAL;
    return r->impl.feed_packed(packed, n, samples_per_buffer, nbuffers, cb, user);
}
extern "C" int adsb_amd_cpr_nl(double lat) { return adsb_amd::cpr_nl(lat); }
extern "C" int adsb_amd_cpr_global(double even_lat, double even_lon, double odd_lat, double odd_lon, int use_even, int32_t* lat1e7, int32_t* lon1e7)
{
    return adsb_amd::cpr_global((int32_t)even_lat, (int32_t)even_lon, (int32_t)odd_lat, (int32_t)odd_lon, use_even != 0, lat1e7, lon1e7) ? 1 : 0; // raw 17-bit values
}
extern "C" void adsb_amd_cpr_global_batch(size_t n, const int32_t* even_lat, const int32_t* even_lon, const int32_t* odd_lat, const int32_t* odd_lon,
                                          const uint8_t* use_even, int32_t* lat1e7, int32_t* lon1e7, uint8_t* ok)
{
    adsb_amd::cpr_global_batch(n, even_lat, even_lon, odd_lat, odd_lon, use_even, lat1e7, lon1e7, ok);
}
extern "C" void adsb_amd_decode_record_host(const adsb_amd_record_t* record, adsb_amd_decoded_t* out)
{
    if (record && out) *out = adsb_amd::decode_record(record->msg, record->df);
}
extern "C" int adsb_amd_decode_1090(adsb_amd_ctx_t* c, const adsb_amd_record_t* records_host, size_t n, adsb_amd_decoded_t* out_host)
{
    if (!c) return ADSB_AMD_EINVAL;
    if (n == 0) return ADSB_AMD_OK;
    if (!records_host || !out_host) return fail(c, ADSB_AMD_EINVAL, "NULL array");
    HIP_TRY(c, hipSetDevice(c->device));
    adsb_amd_record_t*  d_rec = nullptr;
    adsb_amd_decoded_t* d_out = nullptr;
    hipError_t          e     = hipMalloc(&d_rec, n * sizeof(adsb_amd_record_t));
    if (e == hipSuccess) e = hipMalloc(&d_out, n * sizeof(adsb_amd_decoded_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d_rec, records_host, n * sizeof(adsb_amd_record_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_decode1090(d_rec, d_out, n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, d_out, n * sizeof(adsb_amd_decoded_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d_rec) (void)hipFree(d_rec);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess)
    {
        c->error = std::string("decode_1090: ") + hipGetErrorString(e);
        return ADSB_AMD_EHIP;
    }
    return ADSB_AMD_OK;
}
/* A listener that only counts: *(uint64_t*)user += 1 per accepted frame.  For rate measurements through the callback path
 * without a foreign-language trampoline in the loop. */
extern "C" void adsb_amd_count_callback(void* user, const adsb_amd_frame_t* /*frame*/, const adsb_amd_aircraft_t* /*aircraft*/)
{
    if (user) ++*static_cast<uint64_t*>(user);
}
extern "C" size_t adsb_amd_resolver_aircraft_count(const adsb_amd_resolver_t* r) { return r ? r->impl.aircraft_count() : 0; }

// ------------------------------------------------------------------------------------------------ transport
struct adsb_amd_transport final : adsb_amd::Transport::Sink
{
    adsb_amd_transport(const char* path, bool loop) : impl(path ? path : "", loop) {}
    void Deliver(const uint8_t* data, size_t nbytes) override
    {
        if (sink) sink(user, data, nbytes);
    }
    adsb_amd::Transport impl;
    adsb_amd_buffer_fn  sink = nullptr;
    void*               user = nullptr;
};

extern "C" int adsb_amd_transport_create(adsb_amd_transport_t** out, const char* replay_path, int loop)
{
    if (!out) return ADSB_AMD_EINVAL;
    *out = nullptr;
    try
    {
        *out = new adsb_amd_transport(replay_path, loop != 0);
    } catch (...)
    {
        return ADSB_AMD_EHIP;
    }
    return ADSB_AMD_OK;
}
extern "C" void adsb_amd_transport_destroy(adsb_amd_transport_t* t) { delete t; }
extern "C" int  adsb_amd_transport_start(adsb_amd_transport_t* t, adsb_amd_buffer_fn sink, void* user)
{
    if (!t || !sink) return ADSB_AMD_EINVAL;
    t->sink = sink;
    t->user = user;
    try
    {
        t->impl.Start(t);
    } catch (...)
    {
        return ADSB_AMD_EINVAL;
    }
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_transport_stop(adsb_amd_transport_t* t)
{
    if (!t) return ADSB_AMD_EINVAL;
    t->impl.Stop();
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_transport_push(adsb_amd_transport_t* t, const uint8_t* data, size_t nbytes)
{
    if (!t || (!data && nbytes)) return ADSB_AMD_EINVAL;
    try
    {
        t->impl.Push(data, nbytes);
    } catch (...)
    {
        return ADSB_AMD_EINVAL;
    }
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_transport_stats(const adsb_amd_transport_t* t, uint64_t* delivered, int* producer_done, int* page_locked)
{
    if (!t) return ADSB_AMD_EINVAL;
    if (delivered) *delivered = t->impl.Delivered();
    if (producer_done) *producer_done = t->impl.ProducerDone() ? 1 : 0;
    if (page_locked) *page_locked = t->impl.PageLocked() ? 1 : 0;
    return ADSB_AMD_OK;
}

extern "C" long adsb_amd_handler_run_replay(adsb_amd_handler_t* h, const char* path, adsb_amd_on_changed_fn cb, void* user, uint64_t* buffers,
                                            double* seconds)
{
    if (!h || !path) return ADSB_AMD_EINVAL;
    struct stat st;
    if (stat(path, &st) != 0)
    {
        h->error = std::string("cannot open ") + path + ": " + strerror(errno);
        return ADSB_AMD_EINVAL;
    }
    const uint64_t want = (uint64_t)st.st_size / adsb_amd::Transport::kBufferLength;
    struct Sink final : adsb_amd::Transport::Sink
    {
        adsb_amd_handler_t*    h;
        adsb_amd_on_changed_fn cb;
        void*                  user;
        std::atomic<long>      accepted{0}, failed{0}; // written by the transport's consumer thread, polled by the caller's
        void Deliver(const uint8_t* data, size_t nbytes) override
        { // RTLSDR::ConsumerThreadLoop -> IDataHandler::HandleData: one call, one independent buffer
            const long rc = adsb_amd_handler_handle_data(h, data, nbytes, 0, cb, user);
            if (rc < 0) failed.store(rc);
            else accepted.fetch_add(rc);
        }
    } sink;
    sink.h = h, sink.cb = cb, sink.user = user;
    h->error.clear();
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t   got = 0;
    try
    {
        adsb_amd::Transport tr(path, false);
        tr.Start(&sink);
        // until every whole buffer of the file has been delivered -- or the producer has stopped short of that (the file shrank after
        // the stat above, a read failed) and what it queued has been delivered: never wait for buffers that will not come
        while (tr.Delivered() < want && !sink.failed.load() && !tr.Drained()) std::this_thread::sleep_for(std::chrono::microseconds(200));
        tr.Stop();
        got = tr.Delivered();
        if (buffers) *buffers = got;
    } catch (const std::exception& e)
    {
        h->error = e.what();
        return ADSB_AMD_EINVAL;
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (sink.failed.load()) return sink.failed.load();
    if (got < want)
    {
        h->error = std::string(path) + ": the replay ended after " + std::to_string(got) + " of " + std::to_string(want) + " buffers (file truncated or unreadable)";
        return ADSB_AMD_EINVAL;
    }
    return sink.accepted.load();
}
