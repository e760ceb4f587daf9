// resolver1090.cpp -- see resolver1090.hpp.  Pure host C++ (no HIP), so the sequential half is testable without a GPU.
#include "resolver1090.hpp"

#include "decode1090.h"

#include <chrono>
#include <cmath>
#include <cstring>

namespace adsb_amd
{
namespace
{
constexpr int64_t kNsPerSec        = 1000000000LL;
constexpr int64_t kIcaoTtlNs       = 60 * kNsPerSec; // ADSB1090.cpp:202
constexpr int64_t kCprPairSeconds  = 10;             // :1161
} // namespace

// Number of longitude zones, transition latitudes of 1090-WP-9-14 (the table the reference carries at :993-1055).
namespace
{
constexpr double kNlEdge[58] = {
    10.47047130, 14.82817437, 18.18626357, 21.02939493, 23.54504487, 25.82924707, 27.93898710, 29.91135686, 31.77209708,
    33.53993436, 35.22899598, 36.85025108, 38.41241892, 39.92256684, 41.38651832, 42.80914012, 44.19454951, 45.54626723,
    46.86733252, 48.16039128, 49.42776439, 50.67150166, 51.89342469, 53.09516153, 54.27817472, 55.44378444, 56.59318756,
    57.72747354, 58.84763776, 59.95459277, 61.04917774, 62.13216659, 63.20427479, 64.26616523, 65.31845310, 66.36171008,
    67.39646774, 68.42322022, 69.44242631, 70.45451075, 71.45986473, 72.45884545, 73.45177442, 74.43893416, 75.42056257,
    76.39684391, 77.36789461, 78.33374083, 79.29428225, 80.24923213, 81.19801349, 82.13956981, 83.07199445, 83.99173563,
    84.89166191, 85.75541621, 86.53536998, 87.00000000};
// at[q] = number of edges <= q/4 degrees.  Neighbouring edges are at least 0.46 degrees apart, so a quarter degree holds at most
// one: one table look-up and one comparison, no data-dependent branch (the latitudes of a busy sky do not predict).  Built at
// compile time: a function-local static would cost a guard check per call.
struct NlBelow
{
    uint8_t at[4 * 91 + 4];
    double  e[60];
};
constexpr NlBelow make_nl_below()
{
    NlBelow b{};
    for (int k = 0; k < 60; k++) b.e[k] = k < 58 ? kNlEdge[k] : 1e300;
    for (int q = 0; q < 4 * 91 + 4; q++)
    {
        int k = 0;
        while (k < 58 && kNlEdge[k] <= q * 0.25) k++;
        b.at[q] = (uint8_t)k;
    }
    return b;
}
constexpr NlBelow kNlBelow = make_nl_below();
// 360.0 / ni for ni = 1 .. 59: the quotients the reference's expression forms
struct Dlon
{
    double v[60];
};
constexpr Dlon make_dlon()
{
    Dlon d{};
    d.v[0] = 0;
    for (int i = 1; i < 60; i++) d.v[i] = 360.0 / i;
    return d;
}
constexpr Dlon kDlon = make_dlon();
} // namespace

int cpr_nl(double lat)
{
    const double a = lat < 0 ? -lat : lat;
    if (!(a >= 0.0 && a < 91.0))
    { // NaN and out-of-range latitudes: the plain scan
        int k = 0;
        while (k < 58 && !(a < kNlEdge[k])) k++;
        return 59 - k;
    }
    int k = kNlBelow.at[(int)(a * 4.0)];
    k += !(a < kNlBelow.e[k]);
    return 59 - k;
}

namespace
{
int wrap(int a, int b)
{
    const int r = a % b;
    return r < 0 ? r + b : r;
}
// the same for the longitude index, which is almost always in [-b, b): one conditional add, no division
int wrap_small(int a, int b)
{
    const int r = a + (b & (a >> 31));
    return (r >= 0 && r < b) ? r : wrap(a, b);
}
} // namespace

bool cpr_global(int32_t lat0, int32_t lon0, int32_t lat1, int32_t lon1, bool use_even, int32_t* lat1e7, int32_t* lon1e7)
{
    // The reference's expressions (:1079-1121) are in double; with whole-number operands below 2^17 every product, difference and
    // division by 131072 in them is exact, so floor(x / 131072 + 0.5) is the arithmetic shift (x + 65536) >> 17 -- the two zone
    // indices are formed in integers, everything that is scaled by 360/60, 360/59 or 360/ni stays in double as written there.
    const double d0 = 360.0 / 60, d1 = 360.0 / 59;
    const int    j  = (int)((59 * (int64_t)lat0 - 60 * (int64_t)lat1 + 65536) >> 17);
    double       r0 = d0 * (wrap_small(j, 60) + (double)lat0 / 131072);
    double       r1 = d1 * (wrap_small(j, 59) + (double)lat1 / 131072);
    if (r0 >= 270) r0 -= 360;
    if (r1 >= 270) r1 -= 360;
    const int nl0 = cpr_nl(r0), nl1 = cpr_nl(r1);
    if (nl0 != nl1) return false;
    const int    nl  = nl0;
    const int    ni  = use_even ? (nl < 1 ? 1 : nl) : (nl - 1 < 1 ? 1 : nl - 1); // N(lat, 0) / N(lat, 1)
    const int    m   = (int)(((int64_t)lon0 * (nl - 1) - (int64_t)lon1 * nl + 65536) >> 17);
    const double lon = use_even ? lon0 : lon1, rlat = use_even ? r0 : r1;
    double lon_out   = kDlon.v[ni] * (wrap_small(m, ni) + lon / 131072) * 10000000;
    double lat_out   = rlat * 10000000;
    if (lon_out > 180.0 * 10000000) lon_out -= 3600000000.0;
    *lat1e7 = static_cast<int32_t>(lat_out);
    *lon1e7 = static_cast<int32_t>(lon_out);
    return true;
}

void Resolver1090::set_sample_clock(int64_t t0_ns, uint32_t rate_hz)
{
    t0_ns_         = t0_ns;
    rate_hz_       = rate_hz;
    ns_per_sample_ = (rate_hz && 1000000000u % rate_hz == 0) ? 1000000000u / rate_hz : 0;
    rate_recip_    = rate_hz > 1 ? (uint64_t)((((unsigned __int128)1) << 64) / rate_hz) : 0;
}

// InteractiveReceiveData (ADSB1090.cpp:1124-1175) on the decoded fields
void Resolver1090::apply(const adsb_amd_decoded_t& d, int64_t t, Track& a)
{
    // The kind of a frame does not predict (a busy sky interleaves them at random), so the three plain updates are selects, not a
    // switch: measured 1.3 ms of 5.3 per GiB-equivalent of records in the build container went to mispredicted branches here.
    const unsigned k       = d.kind;
    const bool     has_alt = (k == ADSB_AMD_K_ALTITUDE) | (k == ADSB_AMD_K_POSITION);
    a.pub.altitude         = has_alt ? d.altitude : a.pub.altitude;
    uint64_t cs_old;
    std::memcpy(&cs_old, a.pub.callsign, 8);
    const uint64_t cs_new = (uint64_t)d.a | ((uint64_t)d.b << 32);
    const uint64_t cs     = (k == ADSB_AMD_K_IDENT) ? cs_new : cs_old;
    std::memcpy(a.pub.callsign, &cs, 8);
    const bool vel = k == ADSB_AMD_K_VELOCITY;
    a.pub.speed    = vel ? d.a : a.pub.speed;
    a.pub.track    = vel ? d.b : a.pub.track;
    if (k != ADSB_AMD_K_POSITION) return;
    if (d.odd)
    {
        a.odd_lat = (int32_t)d.a;
        a.odd_lon = (int32_t)d.b;
        a.odd_ns  = t;
    }
    else
    {
        a.even_lat = (int32_t)d.a;
        a.even_lon = (int32_t)d.b;
        a.even_ns  = t;
    }
    int64_t whole_seconds = (a.even_ns - a.odd_ns) / kNsPerSec; // duration_cast<seconds>: toward zero
    if (whole_seconds < 0) whole_seconds = -whole_seconds;
    if (whole_seconds <= kCprPairSeconds)
        cpr_global(a.even_lat, a.even_lon, a.odd_lat, a.odd_lon, a.even_ns > a.odd_ns, &a.pub.lat1e7, &a.pub.lon1e7);
}

// Tried and dropped (round 2): splitting a large call into a sequential gating pass on this thread, the per-aircraft updates on
// worker threads (aircraft index modulo the worker count, one CPU each, cache-line-aligned records) and an ordered callback
// pass.  Results identical, but slower everywhere it was measured: 4.0 ms -> 7.2-8.4 ms per GiB of input on the MI355X host
// (EPYC 9575F, 2-16 workers), 1.4 -> 1.6-1.9 ms per 256 MiB in the build container.  The updates are ~10 ns each; handing them
// to another core costs more than doing them (list writes, a cold aircraft line per frame, the wake-up), so the walk stays on
// one thread and what is moved off it is moved to the GPU instead (decode1090.h).
long Resolver1090::feed(const adsb_amd_record_t* rec, const adsb_amd_decoded_t* dec, size_t n, size_t samples_per_buffer, size_t nbuffers,
                        adsb_amd_on_changed_fn cb, void* user)
{
    long     accepted    = 0;
    uint32_t cur_buffer  = 0xFFFFFFFFu;
    uint64_t next_offset = 0; // first offset of the current buffer the reference's loop would still look at
    // Time of a sample = t0 + floor(stream index * 10^9 / rate).  The stream index of a buffer's first sample is split once per
    // buffer into whole seconds and a remainder; inside the buffer only the remainder moves, and its conversion to nanoseconds
    // is a multiplication when 10^9 / rate is whole (2 MS/s), otherwise a multiply-high by floor(2^64 / rate) with the exact
    // fix-up -- no division per frame.  rate 0: wall clock like the reference (:195, :1128, :1161), read once per call; the
    // frames of one call get consecutive nanoseconds so that "the more recent of an even and an odd frame" keeps its order.
    uint64_t      buf_rem = 0;
    int64_t       buf_t   = 0; // t0 + the whole seconds of the buffer's first sample
    const int64_t wall = rate_hz_ ? 0
                                  : std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    for (size_t i = 0; i < n; i++)
    {
        const adsb_amd_record_t& r = rec[i];
        if (r.buffer != cur_buffer)
        {
            cur_buffer  = r.buffer;
            next_offset = 0; // no carry-over between HandleData buffers
            if (rate_hz_)
            {
                const uint64_t first = stream_base_ + static_cast<uint64_t>(r.buffer) * samples_per_buffer;
                buf_t                = t0_ns_ + static_cast<int64_t>(first / rate_hz_) * kNsPerSec;
                buf_rem              = first % rate_hz_;
            }
        }
        if (r.offset < next_offset) continue; // inside a frame that was already accepted (:929-934)
        int64_t t;
        if (rate_hz_)
        {
            uint64_t rem = buf_rem + r.offset;
            t            = buf_t;
            if (rem >= rate_hz_)
            {
                if (rem < 2ull * rate_hz_) rem -= rate_hz_, t += kNsPerSec;
                else t += static_cast<int64_t>(rem / rate_hz_) * kNsPerSec, rem %= rate_hz_;
            }
            uint64_t frac;
            if (ns_per_sample_) frac = rem * ns_per_sample_;
            else
            { // floor(rem * 10^9 / rate) exactly: the estimate is at most two short
                const uint64_t num = rem * 1000000000ull; // rem < rate < 2^32: no overflow
                frac               = (uint64_t)(((unsigned __int128)num * rate_recip_) >> 64);
                uint64_t left      = num - frac * rate_hz_;
                while (left >= rate_hz_) left -= rate_hz_, frac++;
            }
            t += static_cast<int64_t>(frac);
        }
        else t = wall + accepted;
        const bool ap    = (r.flags & ADSB_AMD_F_NEEDS_ICAO) != 0;
        Track*     known = nullptr;
        if (ap)
        { // BruteForceAp: the recovered address must have been seen within the TTL (:200-207, :426)
            known = table_.find(r.addr);
            if (!known || !known->seen || (t - known->seen_ns) > kIcaoTtlNs) continue; // not accepted: the retry record (if any) is next
        }
        bool   created = false;
        Track& a       = known ? *known : table_.get_or_create(r.addr, &created);
        if (created) a.pub.addr = r.addr;
        if (!ap && r.errorbit == -1) a.seen = true, a.seen_ns = t; // clean DF11/17 whitelists its address (:590-594)
        if (dec) apply(dec[i], t, a);
        else apply(decode_record(r.msg, r.df), t, a);
        accepted++;
        // :931 then the loop's j++.  At 2.4 MS/s the frame's (8 + bits) microseconds are (8 + bits) * 2.4 samples, rounded up.
        next_offset = static_cast<uint64_t>(r.offset) + (static_cast<uint64_t>(8 + r.nbits) * per_us_x10_ + 9) / 10 + 1;
        if (cb)
        {
            adsb_amd_frame_t fr{};
            fr.offset = static_cast<uint64_t>(r.buffer) * samples_per_buffer + r.offset;
            std::memcpy(fr.msg, r.msg, 14);
            fr.nbits         = r.nbits;
            fr.errorbit      = r.errorbit;
            fr.pass          = (r.flags & ADSB_AMD_F_PASS2) ? 2 : 1;
            fr.phase_applied = (r.flags & ADSB_AMD_F_PHASE) ? 1 : 0;
            fr.df            = r.df;
            fr.addr          = r.addr;
            cb(user, &fr, &a.pub);
        }
    }
    stream_base_ += static_cast<uint64_t>(samples_per_buffer) * nbuffers;
    return accepted;
}

} // namespace adsb_amd
