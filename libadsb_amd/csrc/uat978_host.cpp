// uat978_host.cpp -- host half of the UAT 978 path and its C ABI (include/adsb_amd.h, "UAT 978" section).
//
//   GPU (uat978.hip)   phases, sign of the phase difference for every sample, every exact 18-bit sync match, and for each
//                      match: the 36-bit sync re-check, the sliced frame and its Reed-Solomon decode, for the match and
//                      for the next sample
//                      ... and which of those frames the dump978 scan loop takes (it jumps over a decoded frame and does not
//                      clear its two shift registers when it does): a successor function over the ordered matches, resolved
//                      by pointer jumping (uat_succ_kernel / uat_mark_kernel)
//   host (this file)   the up-calls for the frames taken, in stream order, and UAT978Handler::HandleData's staging/re-buffering
//                      (UAT978.cpp:43-60).  The scan loop itself is kept here as well (host_loop): it runs when a call needs more
//                      than kUatExtraCap frames behind stale register bits, or when ADSB_AMD_UAT_HOST_LOOP=1 asks for it (tests
//                      hold the two against each other).
//
// The algorithm restated here is the published dump978 legacy demodulator; it is un-vendored in the reference tree
// (SURVEY.md F7), so parity is unpinned: tests compare this path with oracle/oracle978.c on generated streams.
// There is no CPU demodulator in this file: phases, signs, matches, slicing and FEC only ever come from the device.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <condition_variable>
#include <unordered_map>
#include <vector>

#include "adsb_amd.h"
#include "rs978.h"
#include "scan1090.h"
#include "uat978.h"

using namespace adsb_amd;

namespace
{
const RsTables& rs_tables()
{
    static const RsTables t = []
    {
        RsTables x;
        rs978_build_tables(x);
        return x;
    }();
    return t;
}

constexpr uint32_t kCheckMask = (1u << kUatCheckBits) - 1u;
constexpr uint32_t kCheckAdsb = (uint32_t)(0xEACDDA4E2ull >> 18), kCheckUplink = (uint32_t)(0x153225B1Dull >> 18);

// The scan loop's two 18-bit registers are kept here in stream order: bit k = the k-th oldest of the 18 bits (the reference
// shifts left, oldest = most significant; the device delivers sign bits LSB-first).
constexpr uint32_t reverse18(uint32_t x)
{
    uint32_t r = 0;
    for (int k = 0; k < 18; k++) r |= ((x >> k) & 1u) << (17 - k);
    return r;
}
constexpr uint32_t kCheckT[2] = {reverse18(kCheckAdsb), reverse18(kCheckUplink)}; // [0] ADS-B, [1] uplink
// After a jump a register holds (18 - t) bits from before it and t new ones.  If the old content is itself a check word P
// (it is, for the register that fired), the old part can only line up with check word X at these t -- a property of the
// two constants: bit t of kStaleSteps[P][X].  (In fact only P == X, t = 15, 16, 17: the words begin and end with 111 / 000.)
constexpr uint32_t stale_steps(uint32_t oldw, uint32_t x)
{
    uint32_t m = 0;
    for (int t = 1; t <= 17; t++)
        if ((oldw >> t) == (x & ((1u << (18 - t)) - 1u))) m |= 1u << t;
    return m;
}
constexpr uint32_t kStaleSteps[2][2] = {{stale_steps(kCheckT[0], kCheckT[0]), stale_steps(kCheckT[0], kCheckT[1])},
                                        {stale_steps(kCheckT[1], kCheckT[0]), stale_steps(kCheckT[1], kCheckT[1])}};
constexpr uint32_t kAllSteps = 0x3FFFEu; // t = 1 .. 17

// Which of the 17 steps after a jump can fire at all, from two bytes of the register's 35-bit history X = old | new << 18
// (step t looks at bits t .. t + 17 of X): bits 17 .. 24 lie inside the window of every t >= 8, bits 7 .. 14 inside that of
// every t <= 7, so each byte rules out every step whose check word disagrees with it.  Random data leaves a step standing
// with probability 2^-8; the survivors are compared in full.  (Replaces 17 x 4 full comparisons per emitted frame.)
struct StepFilter
{
    uint32_t hi[2][256], lo[2][256]; // [check word][byte] -> steps t consistent with it
};
constexpr StepFilter make_step_filter()
{
    StepFilter f{};
    for (int p = 0; p < 2; p++)
        for (uint32_t v = 0; v < 256; v++)
        {
            uint32_t hi = 0, lo = 0;
            for (int t = 8; t <= 17; t++)
                if (((kCheckT[p] >> (17 - t)) & 0xFFu) == v) hi |= 1u << t;
            for (int t = 1; t <= 7; t++)
                if (((kCheckT[p] >> (7 - t)) & 0xFFu) == v) lo |= 1u << t;
            f.hi[p][v] = hi, f.lo[p][v] = lo;
        }
    return f;
}
constexpr StepFilter kStepFilter = make_step_filter();
inline uint32_t possible_steps(uint32_t oldw, uint32_t fresh)
{
    const uint64_t x  = (uint64_t)oldw | ((uint64_t)fresh << 18);
    const uint32_t b1 = (uint32_t)(x >> 17) & 0xFFu, b0 = (uint32_t)(x >> 7) & 0xFFu;
    return kStepFilter.hi[0][b1] | kStepFilter.lo[0][b0] | kStepFilter.hi[1][b1] | kStepFilter.lo[1][b0];
}

#define UAT_HIP(expr)                                                        \
    do                                                                       \
    {                                                                        \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess)                                                \
        {                                                                    \
            error = std::string(#expr) + ": " + hipGetErrorString(_e);       \
            return ADSB_AMD_EHIP;                                            \
        }                                                                    \
    } while (0)

thread_local std::string g_uat_create_error;

// pinned host array that only grows
template <class T>
struct Pinned
{
    T*     p   = nullptr;
    size_t cap = 0;
    ~Pinned()
    {
        if (p) (void)hipHostFree(p);
    }
    hipError_t reserve(size_t n, size_t keep)
    {
        if (n <= cap) return hipSuccess;
        T*         q = nullptr;
        size_t     c = std::max(n, cap * 2);
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&q), c * sizeof(T));
        if (e != hipSuccess) return e;
        if (keep) std::memcpy(q, p, keep * sizeof(T));
        if (p) (void)hipHostFree(p);
        p = q, cap = c;
        return hipSuccess;
    }
};
} // namespace

struct adsb_amd_uat
{
    int         device = 0;
    uint32_t    ncu    = 256; // compute units of the device (hipDeviceAttributeMultiprocessorCount, read in init)
    hipStream_t stream = nullptr;
    hipEvent_t  ev[6]  = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // What the demodulation pass leaves behind -- records, payloads, uplink payloads, the frames behind frames: 7 MB per GiB -- goes to the host on
    // a stream of its own, beside the decision kernels instead of behind them (decide_and_fetch).
    hipStream_t copy_stream = nullptr;
    hipEvent_t  ev_demod = nullptr, ev_counts = nullptr; // the demodulation pass is done; its counts are on the host
    std::string error;

    uint16_t*             lut_d = nullptr;
    RsTables*             rs_d  = nullptr;
    std::vector<uint16_t> lut_h;

    // scratch sized to the largest stream seen
    uint64_t*  signs_d = nullptr;
    size_t     signs_words = 0;
    uint32_t*  cand_d = nullptr;   // matches as the search flushed them
    uint32_t*  sorted_d = nullptr; // the same in stream order (+ positions the host asked for, appended)
    uint32_t*  order_scratch_d = nullptr;
    size_t     order_scratch_words = 0;
    uint32_t   cand_cap = 0;
    uint32_t*  counts_d = nullptr;
    uint32_t*  demod_work_d = nullptr;
    uint32_t*  counts_h = nullptr; // pinned
    uat_rec_t* recs_d = nullptr;
    uat_win_t* wins_d = nullptr; // parallel to recs_d: the register windows, fetched only for the host's own scan loop
    Pinned<uat_win_t> wins_h;
    bool       wins_on_host = false; // wins_h holds the windows of this call's first nmain records
    uint8_t*   pay_d  = nullptr; // corrected ADS-B frame bytes, kUatPayloadStride per record, parallel to recs_d
    uint8_t*   up_d = nullptr; // decoded uplink payloads, 432 bytes per slot
    uint32_t   up_cap = 0;
    Pinned<uat_rec_t> recs_h;
    Pinned<uint8_t>   pay_h;
    Pinned<uint8_t>   up_h;
    Pinned<uint32_t>  cand_h;
    // the bins of the match search (uat978.h; round 6): two fill-count arrays that swap roles per call, kUatBinCap slots per bin
    uint32_t*  bin_fill_d[2] = {nullptr, nullptr};
    uint32_t*  bin_slots_d = nullptr;
    uint32_t   bins_cap = 0;
    int        bin_phase = 0;
    bool       bins_dirty = false;   // a call ended between the match search and the ordering pass: the current fill counts are not zero
    bool       bins_ordered = false; // the last call's ordering was the one-launch form
    hipStream_t copy_stream2 = nullptr; // the frame bytes beside the records: two copy engines share the host link (46 -> ~55 GB/s)
    // Calls in flight: the three sides' compute streams, made ONE AFTER THE OTHER when the pipeline is first used (side s runs on pipe_stream[s]).
    // The runtime deals streams onto a handful of hardware queues in the order they are made; two sides whose streams share a queue run their kernels
    // one after the other, not side by side (round 6: 0.43 against 0.55 ms per step with the very same kernels, by how many streams the process
    // had made before).  Made together they land on neighbouring queues.  Twins own no stream of their own.
    hipStream_t pipe_stream[3] = {nullptr, nullptr, nullptr};
    hipStream_t own_stream = nullptr; // what `stream` is outside the pipeline (side 0 borrows pipe_stream[0] while it runs a submitted call)
    bool        borrowed_stream = false; // `stream` belongs to the handle that made this twin
    hipEvent_t  ev_copy2 = nullptr;
    std::unordered_map<uint64_t, uint32_t> extra; // positions asked for on top of those: (index << 1 | kind) -> record
    uint32_t   nrecords = 0, nmain = 0, nuplink = 0;
    // the loop's decisions, made on the device (uat978.h: UatArgs from next_bit on)
    uint32_t *   next_bit_d = nullptr, *succ_d = nullptr, *exit_d = nullptr, *emit_d = nullptr, *marks_d = nullptr;
    uat_extra_t* extras_d = nullptr;
    uint8_t*     extra_pay_d = nullptr;
    Pinned<uint32_t>    marks_h;
    Pinned<uat_extra_t> extras_h;
    Pinned<uint8_t>     extra_pay_h;
    uint32_t            nextras = 0;
    bool                decided = false;   // marks_h / extras_h describe this call
    bool                host_loop_only = false;
    uint32_t            extra_cap = kUatExtraCap;
    // a part of a longer stream (adsb_amd_uat_part_scan / _finish): see UatArgs::first_bit.  A whole stream: 1 and "none".
    int64_t             first_bit = 1, end_bit = INT64_MAX;
    const uint16_t*     part_in = nullptr; // the window part_scan left demodulated and undecided
    uint64_t            part_n = 0;
    static constexpr uint32_t kExtraFirstCopy = 128; // extras fetched with the records; a call with more takes a second copy

    // input staging for host buffers
    uint8_t* in_d = nullptr;
    size_t   in_cap = 0;

    // UAT978Handler state (UAT978.cpp:106-109)
    uint16_t* stage_d = nullptr; // 65536 phases
    uint16_t* stage_tmp_d = nullptr;
    size_t    used = 0;
    uint64_t  offset = 0;
    int       carry_full = 0;

    float    scan_ms = 0.f, demod_ms = 0.f;
    float    wall_ms[4] = {0.f, 0.f, 0.f, 0.f}; // last process call: match (launch .. count on the host), demod (.. records on the host), sort, scan loop
    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    uint64_t stat_candidates = 0, stat_extra = 0;

    ~adsb_amd_uat()
    {
        stop_pipeline();
        (void)hipSetDevice(device);
        for (void* p : {(void*)lut_d, (void*)rs_d, (void*)signs_d, (void*)cand_d, (void*)sorted_d, (void*)order_scratch_d, (void*)counts_d, (void*)demod_work_d, (void*)bin_fill_d[0], (void*)bin_fill_d[1], (void*)bin_slots_d, (void*)recs_d, (void*)wins_d, (void*)pay_d, (void*)up_d, (void*)in_d,
                        (void*)stage_d, (void*)stage_tmp_d, (void*)next_bit_d, (void*)succ_d, (void*)exit_d, (void*)emit_d, (void*)marks_d, (void*)extras_d, (void*)extra_pay_d})
            if (p) (void)hipFree(p);
        if (counts_h) (void)hipHostFree(counts_h);
        for (auto e : ev)
            if (e) (void)hipEventDestroy(e);
        if (ev_copy2) (void)hipEventDestroy(ev_copy2);
        if (copy_stream2) (void)hipStreamDestroy(copy_stream2);
        if (ev_demod) (void)hipEventDestroy(ev_demod);
        if (ev_counts) (void)hipEventDestroy(ev_counts);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (auto ps : pipe_stream)
            if (ps) (void)hipStreamDestroy(ps);
        if (borrowed_stream) stream = nullptr;
        if (stream) (void)hipStreamDestroy(stream);
    }

    int init(hipStream_t use_stream = nullptr)
    {
        UAT_HIP(hipSetDevice(device));
        {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) ncu = (uint32_t)n;
        }
        if (use_stream) stream = use_stream, borrowed_stream = true; // a twin: one of the handle's pipeline streams
        else UAT_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        // (the copy streams are made when a call on its own first needs them: decide_and_fetch)
        for (auto& e : ev) UAT_HIP(hipEventCreate(&e));
        UAT_HIP(hipEventCreateWithFlags(&ev_copy2, hipEventDisableTiming));
        UAT_HIP(hipEventCreateWithFlags(&ev_demod, hipEventDisableTiming));
        UAT_HIP(hipEventCreateWithFlags(&ev_counts, hipEventDisableTiming));
        // InitATan2Table, UAT978.cpp:76-100
        lut_h.resize(65536);
        for (unsigned i = 0; i < 256; i++)
            for (unsigned q = 0; q < 256; q++)
            {
                const double ang    = std::atan2((double)q - 127.5, (double)i - 127.5) + M_PI;
                const double scaled = std::round(32768 * ang / M_PI);
                lut_h[i | (q << 8)] = (uint16_t)(scaled < 0 ? 0 : scaled > 65535 ? 65535 : scaled);
            }
        // The demodulation kernel keeps one quadrant of the table in LDS and derives the others (uat978.hip, lut2_folded): that is only
        // the same table if the symmetries hold entry by entry.  They do for this formula (no entry comes within 1e-4 of a rounding tie);
        // a library or compiler whose atan2 breaks them must not demodulate with a different table in silence.
        for (unsigned i = 0; i < 256; i++)
            for (unsigned q = 0; q < 256; q++)
            {
                const uint16_t v = lut_h[i | (q << 8)];
                if ((uint16_t)(v + lut_h[i | ((255 - q) << 8)]) != 0 || (uint16_t)(v + lut_h[(255 - i) | (q << 8)]) != 32768)
                {
                    error = "the phase table lacks the symmetries the demodulation kernel folds it by";
                    return ADSB_AMD_ESTATE;
                }
            }
        UAT_HIP(hipMalloc(&lut_d, 65536 * sizeof(uint16_t)));
        UAT_HIP(hipMemcpy(lut_d, lut_h.data(), 65536 * sizeof(uint16_t), hipMemcpyHostToDevice));
        UAT_HIP(hipMalloc(&rs_d, sizeof(RsTables)));
        UAT_HIP(hipMemcpy(rs_d, &rs_tables(), sizeof(RsTables), hipMemcpyHostToDevice));
        UAT_HIP(hipMalloc(&counts_d, kUatCountWords * sizeof(uint32_t)));
        UAT_HIP(hipMemset(counts_d, 0, kUatCountWords * sizeof(uint32_t)));
        UAT_HIP(hipMalloc(&extras_d, kUatExtraCap * sizeof(uat_extra_t)));
        UAT_HIP(hipMalloc(&extra_pay_d, (size_t)kUatExtraCap * kUatPayloadStride));
        UAT_HIP(extras_h.reserve(kExtraFirstCopy, 0));
        UAT_HIP(extra_pay_h.reserve((size_t)kExtraFirstCopy * kUatPayloadStride, 0));
        {
            const char* v = getenv("ADSB_AMD_UAT_HOST_LOOP");
            if (v && *v && *v != '0') host_loop_only = true;
        }
        UAT_HIP(hipMalloc(&demod_work_d, kUatDemodRanges * 32 * sizeof(uint32_t)));
        UAT_HIP(hipMemset(demod_work_d, 0, kUatDemodRanges * 32 * sizeof(uint32_t)));
        UAT_HIP(hipHostMalloc(&counts_h, kUatCountWords * sizeof(uint32_t))); // uat978.h: UatCount
        UAT_HIP(hipMalloc(&stage_d, 65536 * sizeof(uint16_t)));
        UAT_HIP(hipMalloc(&stage_tmp_d, 65536 * sizeof(uint16_t)));
        UAT_HIP(hipMemset(stage_d, 0, 65536 * sizeof(uint16_t)));
        return ADSB_AMD_OK;
    }

    int fail(int code, const char* msg)
    {
        error = msg;
        return code;
    }

    int reserve_signs(uint64_t nsamples)
    {
        const size_t words = (size_t)((nsamples + 63) / 64) + 2;
        if (words <= signs_words) return ADSB_AMD_OK;
        if (signs_d) (void)hipFree(signs_d);
        signs_d = nullptr, signs_words = 0;
        UAT_HIP(hipMalloc(&signs_d, words * sizeof(uint64_t)));
        signs_words = words;
        return ADSB_AMD_OK;
    }
    // the device arrays only receive; what the host still needs of them has been copied out before they are replaced
    int reserve_cand(uint32_t want)
    {
        if (want <= cand_cap) return ADSB_AMD_OK;
        uint32_t* old_sorted = sorted_d;
        if (cand_d) (void)hipFree(cand_d);
        if (recs_d) (void)hipFree(recs_d);
        if (wins_d) (void)hipFree(wins_d);
        wins_d = nullptr;
        if (pay_d) (void)hipFree(pay_d);
        for (uint32_t** p : {&next_bit_d, &succ_d, &exit_d, &emit_d, &marks_d})
        {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        cand_d = nullptr, recs_d = nullptr, sorted_d = nullptr, pay_d = nullptr;
        const uint32_t old_cap = cand_cap;
        cand_cap = 0;
        UAT_HIP(hipMalloc(&cand_d, (size_t)want * sizeof(uint32_t)));
        UAT_HIP(hipMalloc(&sorted_d, (size_t)want * sizeof(uint32_t)));
        if (old_sorted)
        { // the sorted list is the demod kernel's input and grows by the positions the host asks for: keep it
            UAT_HIP(hipMemcpy(sorted_d, old_sorted, (size_t)std::min(old_cap, want) * sizeof(uint32_t), hipMemcpyDeviceToDevice));
            (void)hipFree(old_sorted);
        }
        UAT_HIP(hipMalloc(&recs_d, (size_t)want * sizeof(uat_rec_t)));
        UAT_HIP(hipMalloc(&wins_d, (size_t)want * sizeof(uat_win_t)));
        UAT_HIP(hipMalloc(&pay_d, (size_t)want * kUatPayloadStride));
        for (uint32_t** p : {&next_bit_d, &succ_d, &exit_d, &emit_d, &marks_d}) UAT_HIP(hipMalloc(p, (size_t)want * sizeof(uint32_t)));
        cand_cap = want;
        return ADSB_AMD_OK;
    }
    int reserve_bins(uint64_t nsamples)
    {
        const uint32_t want = (uint32_t)((nsamples + (1u << kUatBinShift) - 1) >> kUatBinShift);
        if (want <= bins_cap) return ADSB_AMD_OK;
        for (void* q : {(void*)bin_fill_d[0], (void*)bin_fill_d[1], (void*)bin_slots_d})
            if (q) (void)hipFree(q);
        bin_fill_d[0] = bin_fill_d[1] = bin_slots_d = nullptr, bins_cap = 0;
        for (auto& f : bin_fill_d)
        {
            UAT_HIP(hipMalloc(&f, (size_t)want * sizeof(uint32_t)));
            UAT_HIP(hipMemsetAsync(f, 0, (size_t)want * sizeof(uint32_t), stream));
        }
        UAT_HIP(hipMalloc(&bin_slots_d, (size_t)want * kUatBinCap * sizeof(uint32_t)));
        bins_cap = want, bins_dirty = false;
        return ADSB_AMD_OK;
    }
    int reserve_uplink(uint32_t want)
    {
        if (want <= up_cap) return ADSB_AMD_OK;
        if (up_d) (void)hipFree(up_d);
        up_d = nullptr, up_cap = 0;
        UAT_HIP(hipMalloc(&up_d, (size_t)want * 432));
        up_cap = want;
        return ADSB_AMD_OK;
    }

    UatArgs args(const uint16_t* in, uint64_t n, bool phases_given) const
    {
        UatArgs a{};
        a.in = in, a.lut = lut_d, a.rs_tables = rs_d, a.nsamples = n, a.phases_given = phases_given ? 1 : 0, a.ncu = ncu;
        a.signs = signs_d, a.cand = cand_d, a.cand_cap = cand_cap, a.counts = counts_d; // demod_on_device points a.cand at sorted_d
        a.recs = recs_d, a.wins = wins_d, a.payloads = pay_d, a.uplink_payloads = up_d, a.uplink_cap = up_cap, a.demod_work = demod_work_d;
        if (!phases_given) a.bin_fill = bin_fill_d[bin_phase], a.bin_fill_next = bin_fill_d[bin_phase ^ 1], a.bins_cap = bins_cap, a.bin_slots = bin_slots_d;
        a.up_list = order_scratch_d ? order_scratch_d + 2 * (size_t)((n + 32767) / 32768) + 2 : nullptr;
        a.lenbits = (int64_t)(n / 2) - (kUatSyncBits + kUatUplinkBits);
        a.next_bit = next_bit_d, a.extras = extras_d, a.extra_payloads = extra_pay_d, a.extra_cap = extra_cap;
        a.succ = succ_d, a.exit_of = exit_d, a.emit_of = emit_d, a.marks = marks_d;
        a.first_bit = first_bit, a.end_bit = end_bit;
        return a;
    }

    // GPU part of one process_buffer: afterwards recs_h holds one record per exact 18-bit match, in stream order
    // hold_decisions: stop after the demodulation pass (a part of a longer stream: which frames the loop takes in it depends on where
    // the loop stands when it comes in, which the caller learns later -- decide_and_fetch then)
    int scan(const uint16_t* in_dev, uint64_t n, bool phases_given, bool hold_decisions = false)
    {
        if (n >= (1ull << 31)) return fail(ADSB_AMD_EINVAL, "UAT stream longer than 2^31 samples per call");
        part_in = nullptr; // any call takes the buffers a scanned, undecided part was waiting in
        int rc = phases_given ? reserve_signs(n) : reserve_bins(n);
        if (!rc) rc = reserve_cand(std::max<uint32_t>(cand_cap, 4096));
        if (rc) return rc;
        nrecords = nmain = nuplink = nextras = 0;
        wins_on_host = false;
        decided = false;
        extra.clear();
        const double t0 = now_ms();
        for (int attempt = 0;; attempt++)
        {
            if (bins_dirty && !phases_given)
            { // an earlier call (or attempt) filled bins that no ordering pass has taken
                UAT_HIP(hipMemsetAsync(bin_fill_d[bin_phase], 0, (size_t)bins_cap * sizeof(uint32_t), stream));
                bins_dirty = false;
            }
            const UatArgs a = args(in_dev, n, phases_given);
            UAT_HIP(hipEventRecord(ev[0], stream));
            if (!phases_given) bins_dirty = true;
            UAT_HIP(launch_uat978(a, stream));
            UAT_HIP(hipEventRecord(ev[1], stream));
            static_assert(kUatCountMatches == 0 && kUatCountBinOverflow < 8, "one copy of eight words brings both");
            UAT_HIP(hipMemcpyAsync(counts_h, counts_d, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            UAT_HIP(hipStreamSynchronize(stream));
            const uint32_t ncand = counts_h[0];
            if (ncand + 64 > cand_cap)
            { // the match kernel counted past the array (or left no room for later look-ups): grow and repeat the match
                if (attempt > 2) return fail(ADSB_AMD_EHIP, "UAT candidate count keeps changing");
                rc = reserve_cand(ncand + ncand / 4 + 256);
                if (rc) return rc;
                continue;
            }
            stat_candidates += ncand;
            const double t1 = now_ms();
            rc = reserve_uplink(ncand + 64 + kUatExtraCap); // at most one decoded payload per match and per frame behind one
            if (rc) return rc;
            bins_ordered = !phases_given && ncand && counts_h[kUatCountBinOverflow] == 0;
            { // stream order on the device, so that the records arrive in the order the scan loop walks them
                const size_t words = 2 * (size_t)((n + 32767) / 32768) + 2 + cand_cap; // launch_uat978_order: two words per 32 768-sample bin, then the uplink positions
                if (words > order_scratch_words)
                {
                    if (order_scratch_d) (void)hipFree(order_scratch_d);
                    order_scratch_d = nullptr, order_scratch_words = 0;
                    UAT_HIP(hipMalloc(&order_scratch_d, words * sizeof(uint32_t)));
                    order_scratch_words = words;
                }
                if (bins_ordered)
                { // one launch: the match search left the matches binned by position (round 6); the pass zeroes the other fill-count array, the next call's
                    UAT_HIP(launch_uat978_order_bins(args(in_dev, n, phases_given), ncand, sorted_d, stream));
                    bins_dirty = false, bin_phase ^= 1;
                }
                else UAT_HIP(launch_uat978_order(args(in_dev, n, phases_given), ncand, order_scratch_d, sorted_d, stream)); // phases as input (small), or a bin that ran over
            }
            rc = launch_demod(in_dev, n, phases_given, ncand);
            if (!rc && !hold_decisions) rc = decide_and_fetch(in_dev, n, phases_given, ncand);
            if (rc) return rc;
            nmain = ncand;
            UAT_HIP(hipEventElapsedTime(&scan_ms, ev[0], ev[1]));
            const double t2 = now_ms();
            wall_ms[0] = (float)(t1 - t0), wall_ms[1] = (float)(t2 - t1);
            return ADSB_AMD_OK;
        }
    }

    // the demodulation pass over the ordered list
    int launch_demod(const uint16_t* in_dev, uint64_t n, bool phases_given, uint32_t count)
    {
        if (count == 0) return ADSB_AMD_OK;
        const UatArgs a = args(in_dev, n, phases_given);
        UatArgs       d = a;
        d.cand          = sorted_d;
        UAT_HIP(recs_h.reserve(count, 0));
        UAT_HIP(pay_h.reserve((size_t)count * kUatPayloadStride, 0));
        UAT_HIP(hipEventRecord(ev[2], stream));
        // (Round 6 measured the pass cut into four launches over consecutive parts of the list, a part's records on their way to the host beside the
        // next part: 4 x 84 us instead of 197 -- a wave works through thirteen matches of a launch, an uplink match takes ten times an ADS-B one, and a
        // quarter of the list per launch leaves the waves nothing to even that out with.  One launch.)
        UAT_HIP(launch_uat978_demod(d, count, true, stream));
        UAT_HIP(hipEventRecord(ev[3], stream));
        UAT_HIP(hipEventRecord(ev_demod, stream));
        return ADSB_AMD_OK;
    }

    // the loop's decisions, and everything the host needs of the demodulation pass and of them in one wait
    int decide_and_fetch(const uint16_t* in_dev, uint64_t n, bool phases_given, uint32_t count)
    {
        if (count == 0) return ADSB_AMD_OK;
        UatArgs a = args(in_dev, n, phases_given);
        a.cand    = sorted_d;
        const bool     decide = a.lenbits > 0;
        const uint32_t mark_words = (count + 31) / 32;
        if (in_pipeline)
        { // Calls in flight: everything behind the decision kernels, on the call's own stream, one wait.  (The copies of the other form set out
          // earlier but its counts -- sixteen bytes, which the runtime copies with a kernel -- have to get onto a chip that the other calls'
          // scan and demodulation kernels fill, and the worker waits for them: pipelined step 0.44 -> 0.49 ms.)
            if (decide)
            {
                UAT_HIP(marks_h.reserve(mark_words, 0));
                UAT_HIP(hipEventRecord(ev[5], stream));
                UAT_HIP(launch_uat978_decide(a, count, sorted_d, stream, false));
                UAT_HIP(hipEventRecord(ev[4], stream));
                UAT_HIP(hipMemcpyAsync(marks_h.p, marks_d, mark_words * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                UAT_HIP(hipMemcpyAsync(extras_h.p, extras_d, kExtraFirstCopy * sizeof(uat_extra_t), hipMemcpyDeviceToHost, stream));
                UAT_HIP(hipMemcpyAsync(extra_pay_h.p, extra_pay_d, (size_t)kExtraFirstCopy * kUatPayloadStride, hipMemcpyDeviceToHost, stream));
            }
            UAT_HIP(hipMemcpyAsync(counts_h + 1, counts_d + 1, (kUatCountWords - 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            UAT_HIP(hipMemcpyAsync(recs_h.p, recs_d, (size_t)count * sizeof(uat_rec_t), hipMemcpyDeviceToHost, stream));
            UAT_HIP(hipMemcpyAsync(pay_h.p, pay_d, (size_t)count * kUatPayloadStride, hipMemcpyDeviceToHost, stream));
            UAT_HIP(hipStreamSynchronize(stream));
            const uint32_t up_total = counts_h[kUatCountUplinkSlots];
            const bool     overflow = counts_h[kUatCountOverflow] != 0;
            if (up_total > up_cap && !overflow) return fail(ADSB_AMD_EHIP, "UAT uplink payload area too small");
            const uint32_t up_have = std::min(up_total, up_cap);
            nextras                = std::min<uint32_t>(counts_h[kUatCountExtras], extra_cap);
            decided                = decide && !overflow;
            bool more              = false;
            if (up_have > nuplink)
            {
                UAT_HIP(up_h.reserve((size_t)up_have * 432, (size_t)nuplink * 432));
                UAT_HIP(hipMemcpyAsync(up_h.p + (size_t)nuplink * 432, up_d + (size_t)nuplink * 432, (size_t)(up_have - nuplink) * 432,
                                       hipMemcpyDeviceToHost, stream));
                nuplink = up_have, more = true;
            }
            if (decided && nextras > kExtraFirstCopy)
            {
                UAT_HIP(extras_h.reserve(nextras, kExtraFirstCopy));
                UAT_HIP(extra_pay_h.reserve((size_t)nextras * kUatPayloadStride, (size_t)kExtraFirstCopy * kUatPayloadStride));
                UAT_HIP(hipMemcpyAsync(extras_h.p + kExtraFirstCopy, extras_d + kExtraFirstCopy, (size_t)(nextras - kExtraFirstCopy) * sizeof(uat_extra_t),
                                       hipMemcpyDeviceToHost, stream));
                UAT_HIP(hipMemcpyAsync(extra_pay_h.p + (size_t)kExtraFirstCopy * kUatPayloadStride, extra_pay_d + (size_t)kExtraFirstCopy * kUatPayloadStride,
                                       (size_t)(nextras - kExtraFirstCopy) * kUatPayloadStride, hipMemcpyDeviceToHost, stream));
                more = true;
            }
            if (more) UAT_HIP(hipStreamSynchronize(stream));
        }
        else
        { // One call at a time: what the demodulation pass left behind sets out for the host beside the decision kernels (0.733 -> 0.675 ms per call).
            // (1) behind the demodulation pass, on the copy stream: its counts first (they say how much more there is to fetch), then the records
            // and the payloads.  The decision kernels read none of this and write none of it.
            static_assert(kUatCountUplinkSlots == 1 && kUatCountOverflow == 4 && kUatCountFinalBit == 5 && kUatCountTaken == 6, "the two copies of the counts below");
            if (!copy_stream) UAT_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
            UAT_HIP(hipStreamWaitEvent(copy_stream, ev_demod, 0));
            UAT_HIP(hipMemcpyAsync(counts_h + kUatCountUplinkSlots, counts_d + kUatCountUplinkSlots, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, copy_stream));
            UAT_HIP(hipEventRecord(ev_counts, copy_stream));
            UAT_HIP(hipMemcpyAsync(recs_h.p, recs_d, (size_t)count * sizeof(uat_rec_t), hipMemcpyDeviceToHost, copy_stream));
            // (round 6) the frame bytes on a second copy stream, beside the records.  Made on first use: only a handle that runs calls on its own has one
            // -- the runtime deals streams onto a handful of hardware queues in the order they are made, and three more streams that the handle's two
            // twins (calls in flight) never use shifted the twins' compute streams onto one queue: 0.43 -> 0.55 ms per pipelined step inside bench.py
            if (!copy_stream2) UAT_HIP(hipStreamCreateWithFlags(&copy_stream2, hipStreamNonBlocking));
            UAT_HIP(hipStreamWaitEvent(copy_stream2, ev_demod, 0));
            UAT_HIP(hipMemcpyAsync(pay_h.p, pay_d, (size_t)count * kUatPayloadStride, hipMemcpyDeviceToHost, copy_stream2));
            UAT_HIP(hipEventRecord(ev_copy2, copy_stream2));
            if (decide)
            {
                UAT_HIP(hipMemcpyAsync(extras_h.p, extras_d, kExtraFirstCopy * sizeof(uat_extra_t), hipMemcpyDeviceToHost, copy_stream));
                UAT_HIP(hipMemcpyAsync(extra_pay_h.p, extra_pay_d, (size_t)kExtraFirstCopy * kUatPayloadStride, hipMemcpyDeviceToHost, copy_stream));
                // (2) the decisions, on the call's own stream, beside those copies
                UAT_HIP(marks_h.reserve(mark_words, 0));
                UAT_HIP(hipEventRecord(ev[5], stream));
                UAT_HIP(launch_uat978_decide(a, count, sorted_d, stream, true)); // (1024 threads wide: the chip is this call's)
                UAT_HIP(hipEventRecord(ev[4], stream));
                UAT_HIP(hipMemcpyAsync(marks_h.p, marks_d, mark_words * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                UAT_HIP(hipMemcpyAsync(counts_h + kUatCountFinalBit, counts_d + kUatCountFinalBit, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            }
            // (3) as soon as the counts are here: the uplink payloads and the frames behind frames beyond the first copy, still beside the decisions
            UAT_HIP(hipEventSynchronize(ev_counts));
            const uint32_t up_total = counts_h[kUatCountUplinkSlots];
            const bool     overflow = counts_h[kUatCountOverflow] != 0;
            if (up_total > up_cap && !overflow)
            { // copies and kernels of this call are still on their way into arrays the next call may replace: both streams drained first
                (void)hipStreamSynchronize(copy_stream);
                if (copy_stream2) (void)hipStreamSynchronize(copy_stream2);
                (void)hipStreamSynchronize(stream);
                return fail(ADSB_AMD_EHIP, "UAT uplink payload area too small");
            }
            const uint32_t up_have = std::min(up_total, up_cap);
            nextras                = std::min<uint32_t>(counts_h[kUatCountExtras], extra_cap);
            decided                = decide && !overflow;
            if (up_have > nuplink)
            {
                UAT_HIP(up_h.reserve((size_t)up_have * 432, (size_t)nuplink * 432));
                UAT_HIP(hipMemcpyAsync(up_h.p + (size_t)nuplink * 432, up_d + (size_t)nuplink * 432, (size_t)(up_have - nuplink) * 432,
                                       hipMemcpyDeviceToHost, copy_stream));
                nuplink = up_have;
            }
            if (decided && nextras > kExtraFirstCopy)
            {
                UAT_HIP(hipStreamSynchronize(copy_stream)); // (the first copy is on its way into the arrays that grow here)
                UAT_HIP(extras_h.reserve(nextras, kExtraFirstCopy));
                UAT_HIP(extra_pay_h.reserve((size_t)nextras * kUatPayloadStride, (size_t)kExtraFirstCopy * kUatPayloadStride));
                UAT_HIP(hipMemcpyAsync(extras_h.p + kExtraFirstCopy, extras_d + kExtraFirstCopy, (size_t)(nextras - kExtraFirstCopy) * sizeof(uat_extra_t),
                                       hipMemcpyDeviceToHost, copy_stream));
                UAT_HIP(hipMemcpyAsync(extra_pay_h.p + (size_t)kExtraFirstCopy * kUatPayloadStride, extra_pay_d + (size_t)kExtraFirstCopy * kUatPayloadStride,
                                       (size_t)(nextras - kExtraFirstCopy) * kUatPayloadStride, hipMemcpyDeviceToHost, copy_stream));
            }
            UAT_HIP(hipStreamSynchronize(copy_stream));
            UAT_HIP(hipEventSynchronize(ev_copy2));
            UAT_HIP(hipStreamSynchronize(stream));
        }
        nrecords = count;
        UAT_HIP(hipEventElapsedTime(&demod_ms, ev[2], ev[3]));
        wall_ms[2] = 0.f;
        if (decide) UAT_HIP(hipEventElapsedTime(&wall_ms[2], ev[5], ev[4]));
        return ADSB_AMD_OK;
    }

    // record for (index, kind); asks the device when the scan loop reaches a position that is not an exact match in the
    // stream (possible only through stale register bits right after a jump)
    int record_for(const uint16_t* in_dev, uint64_t n, bool phases_given, uint32_t index, uint32_t kind, uint32_t* out)
    {
        { // the first nmain records are in stream order
            const uat_rec_t* lo = recs_h.p;
            const uat_rec_t* hi = recs_h.p + nmain;
            const uat_rec_t* it = std::lower_bound(lo, hi, index, [](const uat_rec_t& r, uint32_t v) { return r.index < v; });
            if (it != hi && it->index == index && it->kind == kind)
            {
                *out = (uint32_t)(it - lo);
                return ADSB_AMD_OK;
            }
        }
        auto ex = extra.find(((uint64_t)index << 1) | kind);
        if (ex != extra.end())
        {
            *out = ex->second;
            return ADSB_AMD_OK;
        }
        if (nrecords + 1 > cand_cap || nuplink + 1 > up_cap)
        { // replace the device arrays by larger ones (host copies are complete); the slot counter keeps counting
            int            rc      = reserve_cand(cand_cap * 2 + 64);
            if (!rc) rc = reserve_uplink(up_cap * 2 + 64);
            if (rc) return rc;
        }
        // One wave, the match word passed by value, the record straight back: a launch, a 64-byte copy and one wait (round 1 staged the
        // word through a copy, reset the work counters and fetched the payload counter as well: six stream operations, ~50 us a look-up).
        stat_extra++;
        const uint32_t at = nrecords;
        {
            UatArgs a     = args(in_dev, n, phases_given);
            a.cand        = nullptr;
            a.single_word = (index & 0x7FFFFFFFu) | (kind << 31);
            a.recs += at;
            a.wins += at;
            a.payloads += (size_t)at * kUatPayloadStride;
            UAT_HIP(recs_h.reserve((size_t)at + 1, at));
            UAT_HIP(wins_h.reserve((size_t)at + 1, wins_on_host ? at : 0));
            UAT_HIP(pay_h.reserve(((size_t)at + 1) * kUatPayloadStride, (size_t)at * kUatPayloadStride));
            UAT_HIP(launch_uat978_demod(a, 1, false, stream));
            UAT_HIP(hipMemcpyAsync(recs_h.p + at, recs_d + at, sizeof(uat_rec_t), hipMemcpyDeviceToHost, stream));
            UAT_HIP(hipMemcpyAsync(wins_h.p + at, wins_d + at, sizeof(uat_win_t), hipMemcpyDeviceToHost, stream));
            UAT_HIP(hipMemcpyAsync(pay_h.p + (size_t)at * kUatPayloadStride, pay_d + (size_t)at * kUatPayloadStride, kUatPayloadStride, hipMemcpyDeviceToHost, stream));
            UAT_HIP(hipStreamSynchronize(stream));
            const uat_rec_t& r = recs_h.p[at];
            if (r.kind == 1 && r.variant < 2)
            { // its decoded payload sits in the next slot of the side array
                if (r.slot >= up_cap) return fail(ADSB_AMD_EHIP, "UAT uplink payload area too small");
                UAT_HIP(up_h.reserve((size_t)(r.slot + 1) * 432, (size_t)nuplink * 432));
                UAT_HIP(hipMemcpyAsync(up_h.p + (size_t)nuplink * 432, up_d + (size_t)nuplink * 432, (size_t)(r.slot + 1 - nuplink) * 432,
                                       hipMemcpyDeviceToHost, stream));
                UAT_HIP(hipStreamSynchronize(stream));
                nuplink = r.slot + 1;
            }
            nrecords = at + 1;
        }
        extra[((uint64_t)index << 1) | kind] = at;
        *out                                 = at;
        return ADSB_AMD_OK;
    }

    // ---------------------------------------------------------------------------------------------------------------
    // process_buffer: the dump978 scan loop over the device's matches.  `len` samples, returns samples consumed.
    // ---------------------------------------------------------------------------------------------------------------
    struct Attempt
    {
        int            skip = 0, rs = 9999, len = 0, variant = 0;
        const uint8_t* data = nullptr;
    };

    // demod_*_frame at index and index + 1, and the reference's choice between them, happened on the device
    bool attempt(const uat_rec_t& r, Attempt& best) const
    {
        if (r.variant > 1) return false;
        best.skip = r.skip, best.rs = r.rs, best.variant = r.variant;
        if (r.kind)
        {
            best.len  = 432;
            best.data = up_h.p + (size_t)r.slot * 432;
        }
        else
        {
            best.len  = r.skip == kUatShortSkip ? 18 : 34; // correct_adsb_frame: the long code is taken iff the type is not 0, the short one iff it is
            best.data = pay_h.p + (size_t)(&r - recs_h.p) * kUatPayloadStride;
        }
        return true;
    }

    int process(const uint16_t* in_dev, uint64_t len, bool phases_given, uint64_t stream_offset, adsb_amd_uat_frame_fn cb, void* user,
                int64_t* consumed)
    {
        int rc = scan(in_dev, len, phases_given);
        if (rc) return rc;
        return scan_loop(in_dev, len, phases_given, stream_offset, cb, user, consumed);
    }

    // the host half of process(): up-calls for the frames the device marked as taken, in stream order (and the frames behind
    // them), and the number of samples the loop consumed
    int scan_loop(const uint16_t* in_dev, uint64_t len, bool phases_given, uint64_t stream_offset, adsb_amd_uat_frame_fn cb, void* user,
                  int64_t* consumed)
    {
        const int64_t lenbits = (int64_t)(len / 2) - (kUatSyncBits + kUatUplinkBits);
        if (host_loop_only || (nmain && lenbits > 0 && !decided)) return host_loop(in_dev, len, phases_given, stream_offset, cb, user, consumed);
        const double t_loop = now_ms();
        auto emit = [&](const uat_rec_t& r, const uint8_t* adsb_bytes)
        {
            const uint8_t* data = r.kind ? up_h.p + (size_t)r.slot * 432 : adsb_bytes;
            const int      n    = r.kind ? 432 : r.skip == kUatShortSkip ? 18 : 34;
            cb(user, r.kind ? '+' : '-', data, n, r.rs, stream_offset + r.index + (uint64_t)r.variant);
        };
        if (decided)
        {
            // the extras of one call are few: by parent, then in the order their wave met them
            std::vector<uint32_t> order(nextras);
            for (uint32_t x = 0; x < nextras; x++) order[x] = x;
            std::sort(order.begin(), order.end(), [&](uint32_t p, uint32_t q) {
                const uat_extra_t &a = extras_h.p[p], &b = extras_h.p[q];
                return a.parent != b.parent ? a.parent < b.parent : a.seq < b.seq;
            });
            auto taken = [&](uint32_t k) { return (marks_h.p[k >> 5] >> (k & 31u)) & 1u; };
            if (cb)
            {
                size_t         xe    = 0;
                const uint32_t words = (nmain + 31) / 32;
                for (uint32_t w = 0; w < words; w++)
                    for (uint32_t bits = marks_h.p[w]; bits; bits &= bits - 1)
                    {
                        const uint32_t k = w * 32 + (uint32_t)__builtin_ctz(bits);
                        emit(recs_h.p[k], pay_h.p + (size_t)k * kUatPayloadStride);
                        while (xe < order.size() && extras_h.p[order[xe]].parent < k) xe++;
                        for (; xe < order.size() && extras_h.p[order[xe]].parent == k; xe++)
                            if (extras_h.p[order[xe]].rec.variant < 2) emit(extras_h.p[order[xe]].rec, extra_pay_h.p + (size_t)order[xe] * kUatPayloadStride);
                    }
            }
            for (uint32_t x : order)
            { // counted as the host loop counts its look-ups: positions the loop reached that are not in the match list
                const uat_extra_t& e = extras_h.p[x];
                if (!taken(e.parent)) continue;
                const uat_rec_t *lo = recs_h.p, *hi = recs_h.p + nmain;
                const uat_rec_t* it = std::lower_bound(lo, hi, e.rec.index, [](const uat_rec_t& r, uint32_t v) { return r.index < v; });
                if (it == hi || it->index != e.rec.index || it->kind != e.rec.kind) stat_extra++;
            }
        }
        const int64_t final_bit = decided ? (int64_t)counts_h[kUatCountFinalBit] : 0;
        *consumed  = lenbits > 0 ? (std::max(final_bit, lenbits) - kUatCheckBits) * 2 : (int64_t)-2 * kUatCheckBits;
        wall_ms[3] = (float)(now_ms() - t_loop);
        return ADSB_AMD_OK;
    }

    // the dump978 scan loop on the host, over the records scan() left in recs_h (see the top of the file for when it runs)
    int host_loop(const uint16_t* in_dev, uint64_t len, bool phases_given, uint64_t stream_offset, adsb_amd_uat_frame_fn cb, void* user,
                  int64_t* consumed)
    {
        int           rc      = ADSB_AMD_OK;
        const double  t_loop  = now_ms();
        const int64_t lenbits = (int64_t)(len / 2) - (kUatSyncBits + kUatUplinkBits);
        if (!wins_on_host && nmain)
        { // the register windows of the matches: only this loop reads them, so only now do they cross the host link
            UAT_HIP(wins_h.reserve(std::max<size_t>(nmain, nrecords), 0));
            UAT_HIP(hipMemcpyAsync(wins_h.p, wins_d, (size_t)nmain * sizeof(uat_win_t), hipMemcpyDeviceToHost, stream));
            UAT_HIP(hipStreamSynchronize(stream));
        }
        wins_on_host = true;
        auto win_of = [&](const uat_rec_t* r) -> const uat_win_t& { return wins_h.p[r - recs_h.p]; };

        int64_t bit = 0; // next bit the loop will examine
        size_t  pos = 0;
        auto rec_at  = [&](size_t k) -> const uat_rec_t& { return recs_h.p[k]; };
        const size_t nordered = nmain;
        auto emit = [&](const uat_rec_t& r, const Attempt& a)
        {
            if (cb) cb(user, r.kind ? '+' : '-', a.data, a.len, a.rs, stream_offset + r.index + (uint64_t)a.variant);
        };

        while (bit < lenbits)
        {
            // --- registers hold only stream bits: the loop fires exactly at the device's matches
            while (pos < nordered && (int64_t)(rec_at(pos).index >> 1) + 17 < std::max<int64_t>(bit, kUatCheckBits)) pos++;
            if (pos >= nordered) break;
            const int64_t startbit = rec_at(pos).index >> 1;
            if (startbit + 17 >= lenbits) break;
            // matches at this start bit: even/odd sample, ADS-B/uplink word
            const uat_rec_t* m[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; // [kind][alignment]
            size_t           q       = pos;
            for (; q < nordered && (int64_t)(rec_at(q).index >> 1) == startbit; q++)
            {
                const uat_rec_t& r     = rec_at(q);
                m[r.kind][r.index & 1] = &r;
            }
            pos = q;
            bit = startbit + 17; // the loop is at this bit now
            const int        kind = (m[0][0] || m[0][1]) ? 0 : 1; // `else if`: the uplink word is only looked at without an ADS-B match
            const uat_rec_t* r    = m[kind][0] ? m[kind][0] : m[kind][1];
            Attempt          a;
            if (!attempt(*r, a))
            {
                bit++;
                continue;
            }
            emit(*r, a);
            // --- jump: bit = startbit + skip, then the loop's ++.  The registers keep their contents, so for the next 17 bits
            // they mix bits from before the jump with new ones and can fire where the stream itself has no match.
            uint32_t oldw[2] = {(uint32_t)win_of(r).window & kCheckMask, (uint32_t)(win_of(r).window >> 32) & kCheckMask};
            uint64_t fresh   = win_of(r).after;
            bit              = startbit + a.skip + 1;
            for (;;)
            {
                // steps worth looking at: all 17, unless a register holds a check word (then only where the words overlap themselves)
                uint32_t steps = 0;
                for (int g = 0; g < 2; g++)
                    steps |= oldw[g] == kCheckT[0] ? (kStaleSteps[0][0] | kStaleSteps[0][1])
                                                   : oldw[g] == kCheckT[1] ? (kStaleSteps[1][0] | kStaleSteps[1][1]) : kAllSteps;
                const uint32_t newb[2] = {(uint32_t)fresh, (uint32_t)(fresh >> 32)};
                steps &= possible_steps(oldw[0], newb[0]) | possible_steps(oldw[1], newb[1]);
                bool           jumped  = false;
                while (steps)
                {
                    const int t = __builtin_ctz(steps);
                    steps &= steps - 1;
                    if (bit + t - 1 >= lenbits) break; // step t examines bit (bit + t - 1)
                    const uint32_t w0 = ((oldw[0] >> t) | (newb[0] << (18 - t))) & kCheckMask;
                    const uint32_t w1 = ((oldw[1] >> t) | (newb[1] << (18 - t))) & kCheckMask;
                    uint32_t       k2;
                    if (w0 == kCheckT[0] || w1 == kCheckT[0]) k2 = 0;
                    else if (w0 == kCheckT[1] || w1 == kCheckT[1]) k2 = 1;
                    else continue;
                    const int64_t  at_bit = bit + t - 1;
                    const int64_t  sb2    = at_bit - kUatCheckBits + 1;
                    const uint32_t index  = (uint32_t)(sb2 * 2 + (w0 == kCheckT[k2] ? 0 : 1));
                    uint32_t       at     = 0;
                    rc                    = record_for(in_dev, len, phases_given, index, k2, &at);
                    if (rc) return rc;
                    const uat_rec_t& r2 = recs_h.p[at];
                    Attempt          a2;
                    if (!attempt(r2, a2)) continue;
                    emit(r2, a2);
                    oldw[0] = w0, oldw[1] = w1;
                    fresh   = win_of(&r2).after;
                    bit     = sb2 + a2.skip + 1;
                    jumped  = true;
                    break;
                }
                if (!jumped)
                {
                    if (bit < lenbits) bit = std::min<int64_t>(bit + 17, lenbits); // both registers hold 18 new bits again from here on;
                                                                                   // a jump past the end of the scanned part stays where it is
                    break;
                }
            }
        }
        if (bit < lenbits) bit = lenbits; // no further match: the loop runs to the end
        *consumed = lenbits > 0 ? (bit - kUatCheckBits) * 2 : (int64_t)-2 * kUatCheckBits;
        wall_ms[3] = (float)(now_ms() - t_loop);
        return ADSB_AMD_OK;
    }

    // ---------------------------------------------------------------------------------------------------------------
    // A part of a longer stream (a recording cut over several GPUs): the window holds the part's own samples,
    // at least 64 samples before them (the loop can fire at a start bit up to 17 bits before the bit it stands at) and, unless it is
    // the stream's last part, three maximum frames and 64 samples behind them (a frame that starts in the part, its stale-register
    // window, a frame behind that and what the demodulating wave reads past a frame's end).  part_scan does everything that does not depend on where the loop stands when it comes into
    // the part; part_finish takes that bit (0 for the first part, the previous part's exit bit otherwise, both counted from the
    // window's first sample), decides, and makes the up-calls for the frames whose start bit is the part's own.
    // ---------------------------------------------------------------------------------------------------------------
    int part_scan(const uint16_t* in_dev, uint64_t n)
    {
        part_in   = nullptr;
        first_bit = 1, end_bit = INT64_MAX;
        const int rc = scan(in_dev, n, false, true);
        if (rc == ADSB_AMD_OK) part_in = in_dev, part_n = n;
        return rc;
    }
    int part_finish(int64_t own_begin_sample, int64_t own_end_sample, int64_t entry_bit, bool last, uint64_t stream_offset, adsb_amd_uat_frame_fn cb,
                    void* user, int64_t* exit_bit, int64_t* consumed)
    {
        if (!part_in) return fail(ADSB_AMD_ESTATE, "adsb_amd_uat_part_finish without a successful adsb_amd_uat_part_scan");
        const uint16_t* in = part_in;
        part_in            = nullptr;
        if (host_loop_only) return fail(ADSB_AMD_ESTATE, "a part of a stream is decided on the device only (host loop selected)");
        const int64_t frame = 2 * (kUatSyncBits + kUatUplinkBits);
        if (own_begin_sample < 0 || own_end_sample < own_begin_sample || (uint64_t)own_end_sample > part_n || entry_bit < 0 ||
            (own_begin_sample != 0 && own_begin_sample < 64) || (!last && (uint64_t)(own_end_sample + 3 * frame + 64) > part_n))
            return fail(ADSB_AMD_EINVAL, "part: the window needs 64 samples before the part's own and, unless it is the last, three maximum frames + 64 behind them");
        first_bit = std::max<int64_t>(entry_bit - (kUatCheckBits - 1), std::max<int64_t>(own_begin_sample / 2, 1));
        end_bit   = last ? INT64_MAX : own_end_sample / 2;
        int rc    = decide_and_fetch(in, part_n, false, nmain);
        first_bit = 1, end_bit = INT64_MAX;
        if (rc) return rc;
        const int64_t lenbits = (int64_t)(part_n / 2) - (kUatSyncBits + kUatUplinkBits);
        if (nmain && lenbits > 0 && !decided) return fail(ADSB_AMD_ENOSPC, "part: more frames behind stale register bits than the device keeps");
        if (!last)
            for (uint32_t x = 0; x < nextras; x++)
            { // a frame behind a frame behind a frame ... that leaves the window was sliced from samples that are not there
                const uat_extra_t& e = extras_h.p[x];
                if (((marks_h.p[e.parent >> 5] >> (e.parent & 31u)) & 1u) && (uint64_t)e.rec.index + (uint64_t)frame + 128 > part_n)
                    return fail(ADSB_AMD_ENOSPC, "part: a chain of frames behind stale register bits leaves the window (give it a longer tail)");
            }
        rc = scan_loop(in, part_n, false, stream_offset, cb, user, consumed);
        if (rc) return rc;
        *exit_bit = std::max<int64_t>(entry_bit, decided ? (int64_t)counts_h[kUatCountFinalBit] : 0);
        return ADSB_AMD_OK;
    }

    // ---------------------------------------------------------------------------------------------------------------
    // Up to three calls in flight on device-resident input: the GPU halves (scan, ordering, demodulation, records to the host)
    // of calls k + 1 and k + 2 run on worker threads while the caller's thread walks the records of call k.  Every call owns
    // a full set of buffers and a stream: this object and its two twins take the calls in turn.  (Two sides leave the GPU idle
    // while the caller is between a collect and the next submit: measured 0.98 ms per GiB step against 1.67 serial.)
    // ---------------------------------------------------------------------------------------------------------------
    struct Job
    {
        const uint16_t* in = nullptr;
        uint64_t        n = 0, offset = 0;
        int             rc = 0;
        bool            queued = false, done = false;
    };
    static constexpr int          kSides = 3;
    std::unique_ptr<adsb_amd_uat> twins[kSides - 1];
    std::thread                   workers[kSides]; // one per side, so that the GPU halves of two calls overlap each other too
    std::mutex                    pipe_mu;
    std::condition_variable       pipe_cv;
    Job                           jobs[kSides];    // [0] runs on this object, [s] on twins[s - 1]
    uint64_t                      submitted = 0, collected = 0;
    bool                          pipe_stop = false;
    bool                          in_pipeline = false; // this side is running a submitted call (decide_and_fetch)

    adsb_amd_uat* side(uint64_t k) { return (k % kSides) ? twins[k % kSides - 1].get() : this; }

    // true while submitted calls have not been collected: the synchronous entry points share this handle's buffers and stream with
    // side 0 of the pipeline and must not run then
    bool calls_in_flight() const { return submitted != collected; }

    void worker_loop(int s)
    {
        const bool device_ok = hipSetDevice(device) == hipSuccess; // per-thread state: a worker that could not select the device fails its jobs
        for (;;)
        {
            Job* j = &jobs[s];
            {
                std::unique_lock<std::mutex> lk(pipe_mu);
                pipe_cv.wait(lk, [&] { return pipe_stop || j->queued; });
                if (pipe_stop) return;
            }
            adsb_amd_uat* const sd   = side((uint64_t)s);
            const hipStream_t   keep = sd->stream;
            if (s == 0) sd->stream = pipe_stream[0]; // (side 0 is this handle: its own stream is for calls on their own)
            sd->in_pipeline = true;
            const int rc = device_ok ? sd->scan(j->in, j->n, false) : sd->fail(ADSB_AMD_EHIP, "hipSetDevice failed on a pipeline worker");
            sd->in_pipeline = false;
            sd->stream      = keep;
            {
                std::lock_guard<std::mutex> lk(pipe_mu);
                j->rc = rc, j->queued = false, j->done = true;
            }
            pipe_cv.notify_all();
        }
    }

    int submit(const uint16_t* in_dev, uint64_t n, uint64_t stream_offset)
    {
        if (submitted - collected >= (uint64_t)kSides) return fail(ADSB_AMD_ESTATE, "as many UAT calls as the handle has buffer sets are in flight already: collect one first");
        if (!pipe_stream[0])
        { // the three sides' compute streams, one after the other (see pipe_stream)
            UAT_HIP(hipSetDevice(device));
            for (auto& ps : pipe_stream) UAT_HIP(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
        }
        for (size_t t = 0; t < sizeof(twins) / sizeof(twins[0]); t++)
            if (auto& twin = twins[t]; !twin)
            {
                twin.reset(new adsb_amd_uat());
                twin->device = device;
                twin->host_loop_only = host_loop_only;
                twin->extra_cap = extra_cap;
                const int rc = twin->init(pipe_stream[t + 1]);
                if (rc)
                {
                    error = twin->error;
                    twin.reset();
                    return rc;
                }
            }
        for (int s = 0; s < kSides; s++)
            if (!workers[s].joinable()) workers[s] = std::thread([this, s] { worker_loop(s); });
        {
            std::lock_guard<std::mutex> lk(pipe_mu);
            Job& j = jobs[submitted % kSides];
            j.in = in_dev, j.n = n, j.offset = stream_offset, j.rc = 0, j.done = false, j.queued = true;
            submitted++;
        }
        pipe_cv.notify_all();
        return ADSB_AMD_OK;
    }

    int collect(adsb_amd_uat_frame_fn cb, void* user, int64_t* consumed)
    {
        if (collected == submitted) return fail(ADSB_AMD_ESTATE, "no UAT call in flight");
        Job j;
        {
            std::unique_lock<std::mutex> lk(pipe_mu);
            pipe_cv.wait(lk, [&] { return jobs[collected % kSides].done; });
            j = jobs[collected % kSides];
        }
        adsb_amd_uat* s = side(collected);
        collected++;
        if (j.rc)
        {
            if (s != this) error = s->error;
            return j.rc;
        }
        const int rc = s->scan_loop(j.in, j.n, false, j.offset, cb, user, consumed);
        if (s != this)
        { // statistics and timings are reported through the handle the caller holds
            error = s->error;
            scan_ms = s->scan_ms, demod_ms = s->demod_ms;
            std::memcpy(wall_ms, s->wall_ms, sizeof(wall_ms));
            stat_candidates += s->stat_candidates, stat_extra += s->stat_extra;
            s->stat_candidates = s->stat_extra = 0;
        }
        return rc;
    }

    void stop_pipeline()
    {
        {
            std::lock_guard<std::mutex> lk(pipe_mu);
            pipe_stop = true;
        }
        pipe_cv.notify_all();
        for (auto& w : workers)
            if (w.joinable()) w.join();
    }

    int upload(const void* host, size_t nbytes)
    {
        if (nbytes > in_cap)
        {
            if (in_d) (void)hipFree(in_d);
            in_d = nullptr, in_cap = 0;
            UAT_HIP(hipMalloc(&in_d, nbytes));
            in_cap = nbytes;
        }
        UAT_HIP(hipMemcpyAsync(in_d, host, nbytes, hipMemcpyHostToDevice, stream));
        return ADSB_AMD_OK;
    }

    // UAT978Handler::HandleData, UAT978.cpp:43-60
    int handle_data(const uint8_t* iq, size_t nbytes, adsb_amd_uat_frame_fn cb, void* user)
    {
        const size_t n = nbytes / 2;
        if (n == 0) return ADSB_AMD_OK;
        int rc = upload(iq, n * 2);
        if (rc) return rc;
        size_t j = 0;
        while (j < n)
        {
            const size_t take = std::min<size_t>(65536 - used, n - j);
            UAT_HIP(launch_phase978(in_d + 2 * j, stage_d + used, take, lut_d, stream));
            const size_t i = used + take;
            j += take;
            int64_t done = 0;
            rc           = process(stage_d, i, true, offset, cb, user, &done);
            if (rc) return rc;
            if (done < 0) done = 0; // fewer than 2 * (36 + 4416) + 2 staged samples: see oracle978_handle_data
            offset += (uint64_t)done;
            // :57 -- the tail length in entries is passed as the byte count; `carry_full` moves the whole tail instead
            const size_t tail  = i - (size_t)done;
            const size_t bytes = carry_full ? tail * sizeof(uint16_t) : tail;
            if (bytes && done)
            {
                UAT_HIP(hipMemcpyAsync(stage_tmp_d, reinterpret_cast<const uint8_t*>(stage_d) + (size_t)done * sizeof(uint16_t), bytes,
                                       hipMemcpyDeviceToDevice, stream));
                UAT_HIP(hipMemcpyAsync(stage_d, stage_tmp_d, bytes, hipMemcpyDeviceToDevice, stream));
            }
            used = tail;
        }
        UAT_HIP(hipStreamSynchronize(stream));
        return ADSB_AMD_OK;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int adsb_amd_uat_create(adsb_amd_uat_t** out, int device)
{
    if (!out) return ADSB_AMD_EINVAL;
    *out = nullptr;
    int        ndev = 0;
    hipError_t e    = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
    {
        g_uat_create_error = e != hipSuccess ? std::string("hipGetDeviceCount: ") + hipGetErrorString(e) : "no such HIP device";
        return ADSB_AMD_ENODEV;
    }
    auto* u = new (std::nothrow) adsb_amd_uat();
    if (!u) return ADSB_AMD_EHIP;
    u->device = device;
    int rc    = u->init();
    if (rc)
    {
        g_uat_create_error = u->error;
        delete u;
        return ADSB_AMD_ENODEV;
    }
    *out = u;
    return ADSB_AMD_OK;
}
extern "C" void        adsb_amd_uat_destroy(adsb_amd_uat_t* u) { delete u; }
extern "C" const char* adsb_amd_uat_last_error(const adsb_amd_uat_t* u) { return u ? u->error.c_str() : g_uat_create_error.c_str(); }
extern "C" int         adsb_amd_uat_set_carry_full(adsb_amd_uat_t* u, int full)
{
    if (!u) return ADSB_AMD_EINVAL;
    u->carry_full = full ? 1 : 0;
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_uat_set_host_loop(adsb_amd_uat_t* u, int on)
{
    if (!u) return ADSB_AMD_EINVAL;
    if (u->calls_in_flight()) return u->fail(ADSB_AMD_ESTATE, "UAT calls are in flight: collect them first");
    u->host_loop_only = on != 0;
    for (auto& t : u->twins)
        if (t) t->host_loop_only = on != 0;
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_uat_set_extra_capacity(adsb_amd_uat_t* u, uint32_t entries)
{
    if (!u || entries > kUatExtraCap) return ADSB_AMD_EINVAL;
    if (u->calls_in_flight()) return u->fail(ADSB_AMD_ESTATE, "UAT calls are in flight: collect them first");
    u->extra_cap = entries;
    for (auto& t : u->twins)
        if (t) t->extra_cap = entries;
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_uat_handle_data(adsb_amd_uat_t* u, const uint8_t* iq_host, size_t nbytes, adsb_amd_uat_frame_fn cb, void* user)
{
    if (!u || (!iq_host && nbytes)) return ADSB_AMD_EINVAL;
    if (u->calls_in_flight()) return u->fail(ADSB_AMD_ESTATE, "UAT calls submitted with adsb_amd_uat_submit_iq are still in flight: collect them first");
    if (hipSetDevice(u->device) != hipSuccess) return u->fail(ADSB_AMD_EHIP, "hipSetDevice failed");
    return u->handle_data(iq_host, nbytes, cb, user);
}
extern "C" int adsb_amd_uat_stream_state(const adsb_amd_uat_t* u, uint64_t* offset, size_t* used)
{
    if (!u) return ADSB_AMD_EINVAL;
    if (offset) *offset = u->offset;
    if (used) *used = u->used;
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_uat_process_phases(adsb_amd_uat_t* u, const uint16_t* phi_host, uint64_t len, uint64_t offset, adsb_amd_uat_frame_fn cb,
                                           void* user, int64_t* consumed)
{
    if (!u || !consumed || (!phi_host && len)) return ADSB_AMD_EINVAL;
    if (u->calls_in_flight()) return u->fail(ADSB_AMD_ESTATE, "UAT calls submitted with adsb_amd_uat_submit_iq are still in flight: collect them first");
    if (hipSetDevice(u->device) != hipSuccess) return u->fail(ADSB_AMD_EHIP, "hipSetDevice failed");
    int rc = u->upload(phi_host, (size_t)len * 2);
    if (rc) return rc;
    return u->process(reinterpret_cast<const uint16_t*>(u->in_d), len, true, offset, cb, user, consumed);
}
extern "C" int adsb_amd_uat_process_iq(adsb_amd_uat_t* u, const void* iq, uint64_t nsamples, int on_device, uint64_t offset,
                                       adsb_amd_uat_frame_fn cb, void* user, int64_t* consumed)
{
    if (!u || !consumed || (!iq && nsamples)) return ADSB_AMD_EINVAL;
    if (u->calls_in_flight()) return u->fail(ADSB_AMD_ESTATE, "UAT calls submitted with adsb_amd_uat_submit_iq are still in flight: collect them first");
    if (hipSetDevice(u->device) != hipSuccess) return u->fail(ADSB_AMD_EHIP, "hipSetDevice failed");
    const uint16_t* dev = reinterpret_cast<const uint16_t*>(iq);
    if (!on_device)
    {
        int rc = u->upload(iq, (size_t)nsamples * 2);
        if (rc) return rc;
        dev = reinterpret_cast<const uint16_t*>(u->in_d);
    }
    else if (reinterpret_cast<uintptr_t>(iq) & 15u) return u->fail(ADSB_AMD_EINVAL, "device IQ pointer must be 16-byte aligned");
    return u->process(dev, nsamples, false, offset, cb, user, consumed);
}
// parity helper (CPU tests): the step filter of the scan loop's post-jump window, and the two check words in stream order
extern "C" uint32_t adsb_amd_uat_possible_steps(uint32_t oldw, uint32_t fresh) { return possible_steps(oldw & kCheckMask, fresh); }
extern "C" uint32_t adsb_amd_uat_check_word(int uplink) { return kCheckT[uplink ? 1 : 0]; }

extern "C" int adsb_amd_uat_submit_iq(adsb_amd_uat_t* u, const void* iq_device, uint64_t nsamples, uint64_t offset)
{
    if (!u || (!iq_device && nsamples)) return ADSB_AMD_EINVAL;
    if (reinterpret_cast<uintptr_t>(iq_device) & 15u) return u->fail(ADSB_AMD_EINVAL, "device IQ pointer must be 16-byte aligned");
    if (hipSetDevice(u->device) != hipSuccess) return u->fail(ADSB_AMD_EHIP, "hipSetDevice failed");
    return u->submit(reinterpret_cast<const uint16_t*>(iq_device), nsamples, offset);
}
extern "C" int adsb_amd_uat_part_scan(adsb_amd_uat_t* u, const void* iq_device, uint64_t nsamples)
{
    if (!u || (!iq_device && nsamples)) return ADSB_AMD_EINVAL;
    if (reinterpret_cast<uintptr_t>(iq_device) & 15u) return u->fail(ADSB_AMD_EINVAL, "device IQ pointer must be 16-byte aligned");
    if (u->calls_in_flight()) return u->fail(ADSB_AMD_ESTATE, "UAT calls submitted with adsb_amd_uat_submit_iq are still in flight: collect them first");
    if (hipSetDevice(u->device) != hipSuccess) return u->fail(ADSB_AMD_EHIP, "hipSetDevice failed");
    return u->part_scan(reinterpret_cast<const uint16_t*>(iq_device), nsamples);
}
extern "C" int adsb_amd_uat_part_finish(adsb_amd_uat_t* u, int64_t own_begin_sample, int64_t own_end_sample, int64_t entry_bit, int last, uint64_t offset,
                                        adsb_amd_uat_frame_fn cb, void* user, int64_t* exit_bit, int64_t* consumed)
{
    if (!u || !exit_bit || !consumed) return ADSB_AMD_EINVAL;
    if (hipSetDevice(u->device) != hipSuccess) return u->fail(ADSB_AMD_EHIP, "hipSetDevice failed");
    return u->part_finish(own_begin_sample, own_end_sample, entry_bit, last != 0, offset, cb, user, exit_bit, consumed);
}
extern "C" int adsb_amd_uat_max_in_flight(void) { return adsb_amd_uat::kSides; }
extern "C" int adsb_amd_uat_collect(adsb_amd_uat_t* u, adsb_amd_uat_frame_fn cb, void* user, int64_t* consumed)
{
    if (!u || !consumed) return ADSB_AMD_EINVAL;
    if (hipSetDevice(u->device) != hipSuccess) return u->fail(ADSB_AMD_EHIP, "hipSetDevice failed");
    return u->collect(cb, user, consumed);
}
extern "C" int adsb_amd_uat_timing(const adsb_amd_uat_t* u, float* scan_ms, float* demod_ms, uint64_t* candidates, uint64_t* extra_lookups)
{
    if (!u) return ADSB_AMD_EINVAL;
    if (scan_ms) *scan_ms = u->scan_ms;
    if (demod_ms) *demod_ms = u->demod_ms;
    if (candidates) *candidates = u->stat_candidates;
    if (extra_lookups) *extra_lookups = u->stat_extra;
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_uat_host_timing(const adsb_amd_uat_t* u, float* match_ms, float* demod_ms, float* sort_ms, float* loop_ms)
{
    if (!u) return ADSB_AMD_EINVAL;
    if (match_ms) *match_ms = u->wall_ms[0];
    if (demod_ms) *demod_ms = u->wall_ms[1];
    if (sort_ms) *sort_ms = u->wall_ms[2];
    if (loop_ms) *loop_ms = u->wall_ms[3];
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_uat_phase_lut(const adsb_amd_uat_t* u, uint16_t* lut65536)
{
    if (!u || !lut65536) return ADSB_AMD_EINVAL;
    std::memcpy(lut65536, u->lut_h.data(), 65536 * sizeof(uint16_t));
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_uat_rs_decode_device(adsb_amd_uat_t* u, int kind, uint8_t* codewords, int count, int* results)
{ // the decoder the demod kernel runs (whole wave on one code word), on `count` words of 30 / 48 / 92 bytes
    if (!u || kind < 0 || kind > 2 || count < 0 || (count && (!codewords || !results))) return ADSB_AMD_EINVAL;
    if (count == 0) return ADSB_AMD_OK;
    if (hipSetDevice(u->device) != hipSuccess) return u->fail(ADSB_AMD_EHIP, "hipSetDevice failed");
    const size_t bytes = (size_t)count * (kind == 0 ? 30 : kind == 1 ? 48 : 92);
    uint8_t*     w_d   = nullptr;
    int*         r_d   = nullptr;
    auto&        error = u->error;
    UAT_HIP(hipMalloc(&w_d, bytes));
    UAT_HIP(hipMalloc(&r_d, (size_t)count * sizeof(int)));
    hipError_t e = hipMemcpy(w_d, codewords, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_uat978_rs_selftest(u->rs_d, kind, w_d, r_d, count, u->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(u->stream);
    if (e == hipSuccess) e = hipMemcpy(codewords, w_d, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(results, r_d, (size_t)count * sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(w_d), (void)hipFree(r_d);
    UAT_HIP(e);
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_uat_rs_decode(int kind, uint8_t* codeword)
{ // the decoder the device runs (rs978.h), compiled for the host: what the CPU tests hold against the oracle
    RsWork w;
    return kind == 0 ? rs978_decode(rs_tables(), 12, 225, codeword, 1, w)
                     : kind == 1 ? rs978_decode(rs_tables(), 14, 207, codeword, 1, w) : rs978_decode(rs_tables(), 20, 163, codeword, 1, w);
}

// ---- the reference's C seam (UAT978.cpp:9-10): same names, same meaning.  dump_raw_message is the host's
// (uat2json-wrapper.cpp:14); it is bound weakly so that this library also loads where no host defines it.
extern "C" void dump_raw_message(char updown, uint8_t* data, int len, int rs_errors) __attribute__((weak));

namespace
{
std::mutex                    g_seam_mutex;
std::unique_ptr<adsb_amd_uat> g_seam;
adsb_amd_dump_raw_message_fn  g_seam_dump = nullptr;

void seam_frame(void*, char updown, const uint8_t* data, int len, int rs_errors, uint64_t)
{
    uint8_t copy[432];
    std::memcpy(copy, data, (size_t)len);
    if (g_seam_dump) g_seam_dump(updown, copy, len, rs_errors);
    else if (dump_raw_message) dump_raw_message(updown, copy, len, rs_errors);
}
} // namespace

extern "C" void adsb_amd_uat_set_dump_raw_message(adsb_amd_dump_raw_message_fn fn) { g_seam_dump = fn; }

extern "C" void init_fec(void)
{
    std::lock_guard<std::mutex> lock(g_seam_mutex);
    if (g_seam) return;
    adsb_amd_uat_t* u = nullptr;
    const char*     d = getenv("ADSB_AMD_DEVICE");
    if (adsb_amd_uat_create(&u, d ? atoi(d) : 0) != ADSB_AMD_OK)
    { // the reference's init_fec cannot fail; a host without a usable GPU must not silently decode nothing
        std::fprintf(stderr, "libadsb_amd: init_fec: %s\n", adsb_amd_uat_last_error(nullptr));
        std::abort();
    }
    g_seam.reset(u);
}

extern "C" int process_buffer(const uint16_t* phi, int len, uint64_t offset)
{
    if (!g_seam) init_fec();
    std::lock_guard<std::mutex> lock(g_seam_mutex);
    int64_t                     consumed = 0;
    int rc = adsb_amd_uat_process_phases(g_seam.get(), phi, len < 0 ? 0 : (uint64_t)len, offset, seam_frame, nullptr, &consumed);
    if (rc != ADSB_AMD_OK)
    {
        std::fprintf(stderr, "libadsb_amd: process_buffer: %s\n", g_seam->error.c_str());
        std::abort();
    }
    return (int)consumed;
}
