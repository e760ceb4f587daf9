// resolver1090.cpp -- see resolver1090.hpp.  Pure host C++ (no HIP), so the sequential half is testable without a GPU.
#include "resolver1090.hpp"

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

// 6-bit AIS character set of the identification message (ADSB1090.cpp:608)
char ais_char(unsigned v)
{
    v &= 63u;
    if (v >= 1 && v <= 26) return static_cast<char>('A' + (v - 1));
    if (v == 32) return ' ';
    if (v >= 48 && v <= 57) return static_cast<char>('0' + (v - 48));
    return '?';
}

// 13-bit AC field with M=0,Q=1 -> feet (:440-466); anything else reports 0
int altitude_ac13(const uint8_t* g)
{
    const bool metric = (g[3] & 0x40) != 0, q = (g[3] & 0x10) != 0;
    if (metric || !q) return 0;
    const int n = ((g[2] & 0x1F) << 6) | ((g[3] & 0x80) >> 2) | ((g[3] & 0x20) >> 1) | (g[3] & 0x0F);
    return n * 25 - 1000;
}
// 12-bit AC field of the airborne position message (:470-486)
int altitude_ac12(const uint8_t* g)
{
    if ((g[5] & 1) == 0) return 0;
    const int n = ((g[5] >> 1) << 4) | (g[6] >> 4);
    return n * 25 - 1000;
}
} // namespace

ModesFields decode_fields(const uint8_t g[14], int df, int nbits, int errorbit, uint32_t icao)
{
    ModesFields f;
    f.df       = df;
    f.nbits    = nbits;
    f.errorbit = errorbit;
    f.icao     = icao;
    f.metype   = g[4] >> 3;
    f.mesub    = g[4] & 7;
    { // Gillham-interleaved identity bits read out as four decimal digits (:560-566)
        const int a = ((g[3] & 0x80) >> 5) | (g[2] & 0x02) | ((g[2] & 0x08) >> 3);
        const int b = ((g[3] & 0x02) << 1) | ((g[3] & 0x08) >> 2) | ((g[3] & 0x20) >> 5);
        const int c = ((g[2] & 0x01) << 2) | ((g[2] & 0x04) >> 1) | ((g[2] & 0x10) >> 4);
        const int d = ((g[3] & 0x01) << 2) | ((g[3] & 0x04) >> 1) | ((g[3] & 0x10) >> 4);
        f.identity  = a * 1000 + b * 100 + c * 10 + d;
    }
    if (df == 0 || df == 4 || df == 16 || df == 20) f.altitude = altitude_ac13(g); // :598
    if (df != 17) return f;

    if (f.metype >= 1 && f.metype <= 4)
    { // eight 6-bit characters in bytes 5..10 (:612-619)
        uint64_t v = 0;
        for (int i = 5; i <= 10; i++) v = (v << 8) | g[i];
        for (int i = 0; i < 8; i++) f.flight[(size_t)i] = ais_char((unsigned)(v >> (42 - 6 * i)));
    }
    else if (f.metype >= 9 && f.metype <= 18)
    { // airborne position (:622-630)
        f.odd      = (g[6] & 0x04) != 0;
        f.altitude = altitude_ac12(g);
        f.raw_lat  = ((g[6] & 3) << 15) | (g[7] << 7) | (g[8] >> 1);
        f.raw_lon  = ((g[8] & 1) << 16) | (g[9] << 8) | g[10];
    }
    else if (f.metype == 19 && f.mesub >= 1 && f.mesub <= 4)
    { // airborne velocity (:631-671)
        if (f.mesub <= 2)
        {
            const int ew = ((g[5] & 3) << 8) | g[6];
            const int ns = ((g[7] & 0x7F) << 3) | (g[8] >> 5);
            f.velocity   = static_cast<int>(std::sqrt(static_cast<double>(ns * ns + ew * ew)));
            if (f.velocity != 0)
            {
                const int    ewv = (g[5] & 4) ? -ew : ew;
                const int    nsv = (g[7] & 0x80) ? -ns : ns;
                const double h   = std::atan2(static_cast<double>(ewv), static_cast<double>(nsv));
                f.heading        = static_cast<int>(h * 360 / (M_PI * 2)); // truncation toward zero, then wrap (:657-659)
                if (f.heading < 0) f.heading += 360;
            }
        }
        else f.heading = static_cast<int>((360.0 / 128) * (((g[5] & 3) << 5) | (g[6] >> 3)));
    }
    return f;
}

// Number of longitude zones, transition latitudes of 1090-WP-9-14 (the table the reference carries at :993-1055).
int cpr_nl(double lat)
{
    static const double edge[58] = {
        10.47047130, 14.82817437, 18.18626357, 21.02939493, 23.54504487, 25.82924707, 27.93898710, 29.91135686, 31.77209708,
        33.53993436, 35.22899598, 36.85025108, 38.41241892, 39.92256684, 41.38651832, 42.80914012, 44.19454951, 45.54626723,
        46.86733252, 48.16039128, 49.42776439, 50.67150166, 51.89342469, 53.09516153, 54.27817472, 55.44378444, 56.59318756,
        57.72747354, 58.84763776, 59.95459277, 61.04917774, 62.13216659, 63.20427479, 64.26616523, 65.31845310, 66.36171008,
        67.39646774, 68.42322022, 69.44242631, 70.45451075, 71.45986473, 72.45884545, 73.45177442, 74.43893416, 75.42056257,
        76.39684391, 77.36789461, 78.33374083, 79.29428225, 80.24923213, 81.19801349, 82.13956981, 83.07199445, 83.99173563,
        84.89166191, 85.75541621, 86.53536998, 87.00000000};
    const double a = lat < 0 ? -lat : lat;
    int          k = 0;
    while (k < 58 && !(a < edge[k])) k++;
    return 59 - k;
}

namespace
{
int wrap(int a, int b)
{
    const int r = a % b;
    return r < 0 ? r + b : r;
}
int zones(double lat, int odd)
{
    const int nl = cpr_nl(lat) - odd;
    return nl < 1 ? 1 : nl;
}
} // namespace

bool cpr_global(double lat0, double lon0, double lat1, double lon1, bool use_even, int32_t* lat1e7, int32_t* lon1e7)
{
    const double d0 = 360.0 / 60, d1 = 360.0 / 59;
    const int    j  = static_cast<int>(std::floor(((59 * lat0 - 60 * lat1) / 131072) + 0.5));
    double       r0 = d0 * (wrap(j, 60) + lat0 / 131072);
    double       r1 = d1 * (wrap(j, 59) + lat1 / 131072);
    if (r0 >= 270) r0 -= 360;
    if (r1 >= 270) r1 -= 360;
    if (cpr_nl(r0) != cpr_nl(r1)) return false;
    double lat_out, lon_out;
    if (use_even)
    {
        const int nl = cpr_nl(r0);
        const int ni = zones(r0, 0);
        const int m  = static_cast<int>(std::floor((((lon0 * (nl - 1)) - (lon1 * nl)) / 131072) + 0.5));
        lon_out      = (360.0 / zones(r0, 0)) * (wrap(m, ni) + lon0 / 131072) * 10000000;
        lat_out      = r0 * 10000000;
    }
    else
    {
        const int nl = cpr_nl(r1);
        const int ni = zones(r1, 1);
        const int m  = static_cast<int>(std::floor((((lon0 * (nl - 1)) - (lon1 * nl)) / 131072.0) + 0.5));
        lon_out      = (360.0 / zones(r1, 1)) * (wrap(m, ni) + lon1 / 131072) * 10000000;
        lat_out      = r1 * 10000000;
    }
    if (lon_out > 180.0 * 10000000) lon_out -= 3600000000.0;
    *lat1e7 = static_cast<int32_t>(lat_out);
    *lon1e7 = static_cast<int32_t>(lon_out);
    return true;
}

void Resolver1090::set_sample_clock(int64_t t0_ns, uint32_t rate_hz)
{
    t0_ns_   = t0_ns;
    rate_hz_ = rate_hz;
}

int64_t Resolver1090::now_ns(uint64_t stream_sample) const
{
    if (rate_hz_ == 0)
        return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    const uint64_t sec = stream_sample / rate_hz_, rem = stream_sample % rate_hz_;
    return t0_ns_ + static_cast<int64_t>(sec) * kNsPerSec + static_cast<int64_t>(rem * 1000000000ULL / rate_hz_);
}

// InteractiveReceiveData (ADSB1090.cpp:1124-1175)
void Resolver1090::apply(const ModesFields& f, int64_t t, Track& a)
{
    if (f.df == 0 || f.df == 4 || f.df == 20)
    {
        a.pub.altitude = f.altitude;
        return;
    }
    if (f.df != 17) return;
    if (f.metype >= 1 && f.metype <= 4) std::memcpy(a.pub.callsign, f.flight.data(), 8);
    else if (f.metype >= 9 && f.metype <= 18)
    {
        a.pub.altitude = f.altitude;
        if (f.odd)
        {
            a.odd_lat = f.raw_lat;
            a.odd_lon = f.raw_lon;
            a.odd_ns  = t;
        }
        else
        {
            a.even_lat = f.raw_lat;
            a.even_lon = f.raw_lon;
            a.even_ns  = t;
        }
        int64_t whole_seconds = (a.even_ns - a.odd_ns) / kNsPerSec; // duration_cast<seconds>: toward zero
        if (whole_seconds < 0) whole_seconds = -whole_seconds;
        if (whole_seconds <= kCprPairSeconds)
            cpr_global(a.even_lat, a.even_lon, a.odd_lat, a.odd_lon, a.even_ns > a.odd_ns, &a.pub.lat1e7, &a.pub.lon1e7);
    }
    else if (f.metype == 19 && (f.mesub == 1 || f.mesub == 2))
    {
        a.pub.speed = static_cast<uint32_t>(f.velocity);
        a.pub.track = static_cast<uint32_t>(f.heading);
    }
}

long Resolver1090::feed(const adsb_amd_record_t* rec, size_t n, size_t samples_per_buffer, size_t nbuffers, adsb_amd_on_changed_fn cb,
                        void* user)
{
    long     accepted    = 0;
    uint32_t cur_buffer  = 0xFFFFFFFFu;
    uint64_t next_offset = 0; // first offset of the current buffer the reference's loop would still look at
    for (size_t i = 0; i < n; i++)
    {
        const adsb_amd_record_t& r = rec[i];
        if (r.buffer != cur_buffer)
        {
            cur_buffer  = r.buffer;
            next_offset = 0; // no carry-over between HandleData buffers
        }
        if (r.offset < next_offset) continue; // inside a frame that was already accepted (:929-934)
        const uint64_t pos = static_cast<uint64_t>(r.buffer) * samples_per_buffer + r.offset;
        const int64_t  t   = now_ns(stream_base_ + pos);
        const bool     ap  = (r.flags & ADSB_AMD_F_NEEDS_ICAO) != 0;
        Track*         known = nullptr;
        if (ap)
        { // BruteForceAp: the recovered address must have been seen within the TTL (:200-207, :426)
            known = table_.find(r.addr);
            if (!known || !known->seen || (t - known->seen_ns) > kIcaoTtlNs) continue; // not accepted: the retry record (if any) is next
        }
        const ModesFields f = decode_fields(r.msg, r.df, r.nbits, r.errorbit, r.addr);
        bool              created = false;
        Track&            a       = known ? *known : table_.get_or_create(r.addr, &created);
        if (created) a.pub.addr = r.addr;
        if (!ap && r.errorbit == -1) a.seen = true, a.seen_ns = t; // clean DF11/17 whitelists its address (:590-594)
        apply(f, t, a);
        accepted++;
        next_offset = static_cast<uint64_t>(r.offset) + static_cast<uint64_t>(8 + r.nbits) * 2 + 1; // :931 then the loop's j++
        if (cb)
        {
            adsb_amd_frame_t fr{};
            fr.offset = pos;
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
