// resolver1090.cpp -- see resolver1090.hpp.  Pure host C++ (no HIP), so the sequential half is testable without a GPU.
#include "resolver1090.hpp"

#include "decode1090.h"

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstring>

#include <immintrin.h>
#include <sched.h>

#include <cstdio>
#include <cstdlib>

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

// ------------------------------------------------------------------------------------------------
// cpr_global for four pairs at a time.  Every double operation below is the scalar function's, in the same order (the build keeps
// -ffp-contract=off, so nothing is fused); divisions by 131072 are exact in both forms; the zone indices are the same integers.
// ------------------------------------------------------------------------------------------------
namespace
{
#if defined(__AVX2__)
inline __m128i wrap4(__m128i a, __m128i b)
{ // a mod b for a in [-2 b, 2 b): what wrap() returns for the zone indices (|j| <= 60, |m| <= nl <= ni + 1)
    const __m128i zero = _mm_setzero_si128();
    a                  = _mm_add_epi32(a, _mm_and_si128(b, _mm_cmpgt_epi32(zero, a)));
    a                  = _mm_add_epi32(a, _mm_and_si128(b, _mm_cmpgt_epi32(zero, a)));
    a                  = _mm_sub_epi32(a, _mm_andnot_si128(_mm_cmpgt_epi32(b, a), b));
    a                  = _mm_sub_epi32(a, _mm_andnot_si128(_mm_cmpgt_epi32(b, a), b));
    return a;
}
inline __m128i nl4(__m256d lat)
{ // cpr_nl: |lat| < 360 here, so the table index is clamped to its last entry, which answers like the plain scan does above 87 degrees
    const __m256d a = _mm256_andnot_pd(_mm256_set1_pd(-0.0), lat);
    __m128i       q = _mm256_cvttpd_epi32(_mm256_mul_pd(a, _mm256_set1_pd(4.0)));
    q               = _mm_min_epi32(q, _mm_set1_epi32(4 * 91 + 3));
    __m128i k       = _mm_and_si128(_mm_i32gather_epi32(reinterpret_cast<const int*>(kNlBelow.at), q, 1), _mm_set1_epi32(0xFF));
    const __m256d e = _mm256_i32gather_pd(kNlBelow.e, k, 8);
    const __m256d lt = _mm256_cmp_pd(a, e, _CMP_LT_OQ); // a < e[k]
    // k += !(a < e[k])
    const __m128i lt32 = _mm256_castsi256_si128(_mm256_permutevar8x32_epi32(_mm256_castpd_si256(lt), _mm256_setr_epi32(0, 2, 4, 6, 0, 0, 0, 0)));
    k                  = _mm_add_epi32(k, _mm_andnot_si128(lt32, _mm_set1_epi32(1)));
    return _mm_sub_epi32(_mm_set1_epi32(59), k);
}
#endif
} // namespace

void cpr_global_batch(size_t n, const int32_t* lat0, const int32_t* lon0, const int32_t* lat1, const int32_t* lon1, const uint8_t* use_even, int32_t* lat1e7,
                      int32_t* lon1e7, uint8_t* ok)
{
    size_t i = 0;
#if defined(__AVX2__)
    static_assert(offsetof(NlBelow, e) >= sizeof(kNlBelow.at), "the byte gather reads three bytes past an entry: into the struct, never past it");
    const __m128i one = _mm_set1_epi32(1), half = _mm_set1_epi32(65536);
    const __m256d inv = _mm256_set1_pd(1.0 / 131072), d0 = _mm256_set1_pd(360.0 / 60), d1 = _mm256_set1_pd(360.0 / 59);
    const __m256d c270 = _mm256_set1_pd(270), c360 = _mm256_set1_pd(360), e7 = _mm256_set1_pd(10000000);
    for (; i + 4 <= n; i += 4)
    {
        const __m128i a0 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(lat0 + i)), o0 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(lon0 + i));
        const __m128i a1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(lat1 + i)), o1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(lon1 + i));
        int32_t       ue4;
        std::memcpy(&ue4, use_even + i, 4);
        const __m128i ue = _mm_cmpgt_epi32(_mm_cvtepu8_epi32(_mm_cvtsi32_si128(ue4)), _mm_setzero_si128()); // all ones where the even frame is the newer one
        // j = (59 lat0 - 60 lat1 + 65536) >> 17 (all operands < 2^23)
        const __m128i j  = _mm_srai_epi32(_mm_add_epi32(_mm_sub_epi32(_mm_mullo_epi32(a0, _mm_set1_epi32(59)), _mm_mullo_epi32(a1, _mm_set1_epi32(60))), half), 17);
        const __m256d f0 = _mm256_mul_pd(_mm256_cvtepi32_pd(a0), inv), f1 = _mm256_mul_pd(_mm256_cvtepi32_pd(a1), inv);
        __m256d       r0 = _mm256_mul_pd(d0, _mm256_add_pd(_mm256_cvtepi32_pd(wrap4(j, _mm_set1_epi32(60))), f0));
        __m256d       r1 = _mm256_mul_pd(d1, _mm256_add_pd(_mm256_cvtepi32_pd(wrap4(j, _mm_set1_epi32(59))), f1));
        r0               = _mm256_sub_pd(r0, _mm256_and_pd(_mm256_cmp_pd(r0, c270, _CMP_GE_OQ), c360));
        r1               = _mm256_sub_pd(r1, _mm256_and_pd(_mm256_cmp_pd(r1, c270, _CMP_GE_OQ), c360));
        const __m128i nl0 = nl4(r0), nl1 = nl4(r1);
        const __m128i good = _mm_cmpeq_epi32(nl0, nl1);
        const __m128i nlm = _mm_sub_epi32(nl0, one);
        // ni = use_even ? max(nl, 1) : max(nl - 1, 1)
        const __m128i ni = _mm_max_epi32(_mm_blendv_epi8(nlm, nl0, ue), one);
        // m = (lon0 (nl - 1) - lon1 nl + 65536) >> 17
        const __m128i m  = _mm_srai_epi32(_mm_add_epi32(_mm_sub_epi32(_mm_mullo_epi32(o0, nlm), _mm_mullo_epi32(o1, nl0)), half), 17);
        const __m128i wm = wrap4(m, ni);
        const __m256d uem = _mm256_castsi256_pd(_mm256_cvtepi32_epi64(ue));
        const __m256d lon = _mm256_blendv_pd(_mm256_cvtepi32_pd(o1), _mm256_cvtepi32_pd(o0), uem);
        const __m256d rlat = _mm256_blendv_pd(r1, r0, uem);
        const __m256d dlon = _mm256_div_pd(c360, _mm256_cvtepi32_pd(ni)); // kDlon.v[ni] is this quotient
        __m256d       lo   = _mm256_mul_pd(_mm256_mul_pd(dlon, _mm256_add_pd(_mm256_cvtepi32_pd(wm), _mm256_mul_pd(lon, inv))), e7);
        const __m256d la   = _mm256_mul_pd(rlat, e7);
        lo = _mm256_sub_pd(lo, _mm256_and_pd(_mm256_cmp_pd(lo, _mm256_set1_pd(180.0 * 10000000), _CMP_GT_OQ), _mm256_set1_pd(3600000000.0)));
        alignas(16) int32_t la4[4], lo4[4], ok4[4];
        _mm_store_si128(reinterpret_cast<__m128i*>(la4), _mm256_cvttpd_epi32(la));
        _mm_store_si128(reinterpret_cast<__m128i*>(lo4), _mm256_cvttpd_epi32(lo));
        _mm_store_si128(reinterpret_cast<__m128i*>(ok4), good);
        for (int k = 0; k < 4; k++)
        {
            ok[i + k] = ok4[k] ? 1 : 0;
            if (ok4[k]) lat1e7[i + k] = la4[k], lon1e7[i + k] = lo4[k];
        }
    }
#endif
    for (; i < n; i++) ok[i] = cpr_global(lat0[i], lon0[i], lat1[i], lon1[i], use_even[i] != 0, &lat1e7[i], &lon1e7[i]) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
struct Resolver1090::Block
{
    static constexpr size_t kFrames = 1024;
    size_t ne = 0, np = 0; // accepted frames / pairs in the block
    // accepted frames of the batch, in order
    uint32_t           rec[kFrames];   // index into the caller's record array
    uint32_t           trk[kFrames];   // aircraft index
    int32_t            pair[kFrames];  // index into the pair arrays below, -1: the frame completes no pair
    adsb_amd_decoded_t dec[kFrames];   // fields decoded on the host (when the caller passed none)
    // even/odd pairs to decode, and the results
    int32_t lat0[kFrames + 4], lon0[kFrames + 4], lat1[kFrames + 4], lon1[kFrames + 4];
    uint8_t even[kFrames + 4];
    int32_t out_lat[kFrames + 4], out_lon[kFrames + 4];
    uint8_t ok[kFrames + 4];
};

Resolver1090::Resolver1090() : blocks_(new Block[kRing])
{
    gates_.reserve(512);
    pubs_.reserve(512);
}
Resolver1090::~Resolver1090()
{
    if (helper_.joinable())
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        helper_.join();
    }
    delete[] blocks_;
}

void Resolver1090::set_sample_clock(int64_t t0_ns, uint32_t rate_hz)
{
    t0_ns_         = t0_ns;
    rate_hz_       = rate_hz;
    ns_per_sample_ = (rate_hz && 1000000000u % rate_hz == 0) ? 1000000000u / rate_hz : 0;
    rate_recip_    = rate_hz > 1 ? (uint64_t)((((unsigned __int128)1) << 64) / rate_hz) : 0;
}

// ---------------- pass 1: which records the reference accepts, and which even/odd pair each position frame completes.  Touches the
// address table and the gate records only (new aircraft get their index here; their published record is created by the update pass).
// SRC: 0 records + the GPU's decoded fields, 1 records decoded here, 2 the packed form.  A packed entry starts like a record (buffer,
// offset, addr, reserved, nbits, errorbit, df, flags at the same offsets) and ends with kind, odd, altitude, a, b.
namespace
{
struct Head // the first 18 bytes of adsb_amd_record_t and of adsb_amd_packed_t
{
    uint32_t buffer, offset, addr;
    uint16_t reserved;
    uint8_t  nbits;
    int8_t   errorbit;
    uint8_t  df, flags;
};
static_assert(offsetof(adsb_amd_record_t, df) == 16 && offsetof(adsb_amd_packed_t, df) == 16 && offsetof(adsb_amd_record_t, flags) == 17 &&
                  offsetof(adsb_amd_packed_t, flags) == 17 && offsetof(adsb_amd_packed_t, kind) == 18 && sizeof(adsb_amd_packed_t) == 32 &&
                  sizeof(adsb_amd_record_t) == 32 && offsetof(adsb_amd_record_t, errorbit) == offsetof(adsb_amd_packed_t, errorbit),
              "record and packed entry share their head");
inline const Head& head_of(const void* base, size_t i) { return *reinterpret_cast<const Head*>(static_cast<const uint8_t*>(base) + 32 * i); }
} // namespace

template <int SRC>
void Resolver1090::gate_pass(Block& blk, Walk& w, const Job& job)
{
    constexpr bool            HOST_DECODE = SRC == 1;
    const void*               heads = SRC == 2 ? static_cast<const void*>(job.pk) : static_cast<const void*>(job.rec);
    const adsb_amd_record_t*  rec = job.rec;
    const adsb_amd_decoded_t* dec = job.dec;
    const adsb_amd_packed_t*  pk  = job.pk;
    const size_t              n   = job.n;
    // Time of a sample = t0 + floor(stream index * 10^9 / rate).  The stream index of a buffer's first sample is split once per
    // buffer into whole seconds and a remainder; inside the buffer only the remainder moves, and its conversion to nanoseconds
    // is a multiplication when 10^9 / rate is whole (2 MS/s), otherwise a multiply-high by floor(2^64 / rate) with the exact
    // fix-up -- no division per frame.  rate 0: wall clock like the reference (:195, :1128, :1161), read once per call; the
    // frames of one call get consecutive nanoseconds so that "the more recent of an even and an odd frame" keeps its order.
    size_t ne = 0, np = 0;
    size_t i  = w.i;
    for (; i < n && ne < Block::kFrames; i++)
    {
        const Head& r = head_of(heads, i);
        if (r.buffer != w.cur_buffer)
        {
            w.cur_buffer  = r.buffer;
            w.next_offset = 0; // no carry-over between HandleData buffers
            if (rate_hz_)
            {
                const uint64_t first = stream_base_ + static_cast<uint64_t>(r.buffer) * job.samples_per_buffer;
                w.buf_t              = t0_ns_ + static_cast<int64_t>(first / rate_hz_) * kNsPerSec;
                w.buf_rem            = first % rate_hz_;
            }
        }
        if (r.offset < w.next_offset) continue; // inside a frame that was already accepted (:929-934)
        int64_t t;
        if (rate_hz_)
        {
            uint64_t rem = w.buf_rem + r.offset;
            t            = w.buf_t;
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
        else t = w.wall + w.accepted + (long)ne;
        const bool ap  = (r.flags & ADSB_AMD_F_NEEDS_ICAO) != 0;
        int32_t    idx = table_.find(r.addr);
        if (ap)
        { // BruteForceAp: the recovered address must have been seen within the TTL (:200-207, :426)
            if (idx < 0) continue; // not accepted: the retry record (if any) is next
            const Gate& k = gates_[(size_t)idx];
            if (!k.seen || (t - k.seen_ns) > kIcaoTtlNs) continue;
        }
        else if (idx < 0)
        {
            idx = (int32_t)table_.insert(r.addr);
            gates_.emplace_back();
        }
        Gate&      g     = gates_[(size_t)idx];
        const bool clean = !ap && r.errorbit == -1; // clean DF11/17 whitelists its address (:590-594)
        g.seen |= clean;
        g.seen_ns = clean ? t : g.seen_ns;
        if (HOST_DECODE) blk.dec[ne] = decode_record(rec[i].msg, r.df);
        struct Fields
        {
            unsigned kind, odd;
            uint32_t a, b;
        };
        const Fields d = SRC == 2   ? Fields{pk[i].kind, pk[i].odd, pk[i].a, pk[i].b}
                         : SRC == 1 ? Fields{blk.dec[ne].kind, blk.dec[ne].odd, blk.dec[ne].a, blk.dec[ne].b}
                                    : Fields{dec[i].kind, dec[i].odd, dec[i].a, dec[i].b};
        // A position frame replaces its half of the pair; the pair decodes when the halves are at most ten whole seconds apart
        // (:1161: duration_cast<seconds> truncates toward zero, so |even - odd| < 11 s).  Written as selects: the kind of a frame and
        // its format flag do not predict.  The pair's operands are copied out now -- a later frame of the batch may change them.
        const bool     pos = d.kind == ADSB_AMD_K_POSITION;
        const unsigned o   = d.odd & 1u;
        g.lat[o]           = pos ? (int32_t)d.a : g.lat[o];
        g.lon[o]           = pos ? (int32_t)d.b : g.lon[o];
        g.pos_ns[o]        = pos ? t : g.pos_ns[o];
        int64_t apart      = g.pos_ns[0] - g.pos_ns[1];
        apart              = apart < 0 ? -apart : apart;
        const bool within  = pos & (apart < (kCprPairSeconds + 1) * kNsPerSec);
        blk.lat0[np]       = g.lat[0];
        blk.lon0[np]       = g.lon[0];
        blk.lat1[np]       = g.lat[1];
        blk.lon1[np]       = g.lon[1];
        blk.even[np]       = g.pos_ns[0] > g.pos_ns[1];
        blk.rec[ne]        = (uint32_t)i;
        blk.trk[ne]        = (uint32_t)idx;
        blk.pair[ne]       = within ? (int32_t)np : -1;
        np += within;
        ne++;
        // :931 then the loop's j++.  At 2.4 MS/s the frame's (8 + bits) microseconds are (8 + bits) * 2.4 samples, rounded up.
        w.next_offset = static_cast<uint64_t>(r.offset) + (static_cast<uint64_t>(8 + r.nbits) * per_us_x10_ + 9) / 10 + 1;
    }
    w.i = i;
    w.accepted += (long)ne;
    blk.ne = ne;
    blk.np = np;
}

// ---------------- pass 2, the block's pairs (:1079-1121), and pass 3: InteractiveReceiveData (:1124-1175) on the decoded fields, frame by
// frame, and the callback.  Touches the published aircraft records only.
void Resolver1090::update_pass(Block& blk, const Job& job, adsb_amd_on_changed_fn cb, void* user)
{
    cpr_global_batch(blk.np, blk.lat0, blk.lon0, blk.lat1, blk.lon1, blk.even, blk.out_lat, blk.out_lon, blk.ok);
    const bool  packed = job.pk != nullptr, host_decode = !packed && job.dec == nullptr;
    const void* heads  = packed ? static_cast<const void*>(job.pk) : static_cast<const void*>(job.rec);
    for (size_t e = 0; e < blk.ne; e++)
    {
        const size_t i = blk.rec[e];
        const Head&  r = head_of(heads, i);
        struct Fields
        {
            unsigned kind;
            int32_t  altitude;
            uint32_t a, b;
        };
        const Fields d = packed        ? Fields{job.pk[i].kind, job.pk[i].altitude, job.pk[i].a, job.pk[i].b}
                         : host_decode ? Fields{blk.dec[e].kind, blk.dec[e].altitude, blk.dec[e].a, blk.dec[e].b}
                                       : Fields{job.dec[i].kind, job.dec[i].altitude, job.dec[i].a, job.dec[i].b};
        const size_t t = blk.trk[e];
        if (t >= pubs_.size())
        { // the gate pass met this aircraft for the first time in this frame (indices are handed out in order)
            pubs_.resize(t + 1);
            pubs_[t].addr = r.addr;
        }
        adsb_amd_aircraft_t& a = pubs_[t];
        // the kind of a frame does not predict (a busy sky interleaves them at random): selects, not a switch
        const unsigned k       = d.kind;
        const bool     has_alt = (k == ADSB_AMD_K_ALTITUDE) | (k == ADSB_AMD_K_POSITION);
        a.altitude             = has_alt ? d.altitude : a.altitude;
        uint64_t cs_old;
        std::memcpy(&cs_old, a.callsign, 8);
        const uint64_t cs_new = (uint64_t)d.a | ((uint64_t)d.b << 32);
        const uint64_t cs     = (k == ADSB_AMD_K_IDENT) ? cs_new : cs_old;
        std::memcpy(a.callsign, &cs, 8);
        const bool vel = k == ADSB_AMD_K_VELOCITY;
        a.speed        = vel ? d.a : a.speed;
        a.track        = vel ? d.b : a.track;
        const int32_t p   = blk.pair[e];
        const size_t  pi  = p < 0 ? 0 : (size_t)p;
        const bool    fix = (p >= 0) & (blk.ok[pi] != 0);
        a.lat1e7          = fix ? blk.out_lat[pi] : a.lat1e7;
        a.lon1e7          = fix ? blk.out_lon[pi] : a.lon1e7;
        if (cb)
        {
            adsb_amd_frame_t fr{};
            fr.offset = frame_base_ + static_cast<uint64_t>(r.buffer) * job.samples_per_buffer + r.offset;
            if (!packed) std::memcpy(fr.msg, job.rec[i].msg, 14);
            fr.nbits         = r.nbits;
            fr.errorbit      = r.errorbit;
            fr.pass          = (r.flags & ADSB_AMD_F_PASS2) ? 2 : 1;
            fr.phase_applied = (r.flags & ADSB_AMD_F_PHASE) ? 1 : 0;
            fr.df            = r.df;
            fr.addr          = r.addr;
            cb(user, &fr, &a);
        }
    }
}

// The helper: waits for a job, runs the sequential pass over it block by block into the ring, never more than kRing blocks ahead of
// the update pass.
namespace
{
// A CPU that shares its last-level cache with `cpu`, is allowed to this process and is not `cpu`'s own hyper-thread sibling when there
// is another choice; -1: none found.  The two passes hand 20 KB per block to each other: across two L3 domains (two CCDs of an EPYC)
// that transfer costs more than the pipeline gains (measured: 3.9 ms per GiB of input instead of 2.x on the MI355X host).
int cache_neighbour(int cpu)
{
    auto read_list = [](const char* fmt, int c, cpu_set_t* out) {
        char path[128], buf[512];
        std::snprintf(path, sizeof path, fmt, c);
        CPU_ZERO(out);
        FILE* f = std::fopen(path, "r");
        if (!f) return false;
        const bool ok = std::fgets(buf, sizeof buf, f) != nullptr;
        std::fclose(f);
        if (!ok) return false;
        for (char* p = buf; *p && *p != '\n';)
        {
            char*      e;
            const long a = std::strtol(p, &e, 10);
            long       b = a;
            if (e == p) break;
            if (*e == '-') b = std::strtol(e + 1, &e, 10);
            for (long k = a; k <= b && k < CPU_SETSIZE; k++) CPU_SET((int)k, out);
            p = (*e == ',') ? e + 1 : e;
        }
        return true;
    };
    cpu_set_t allowed, l3, smt;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return -1;
    if (!read_list("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu, &l3)) return -1;
    if (!read_list("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu, &smt)) CPU_ZERO(&smt);
    int fallback = -1;
    for (int k = 1; k < CPU_SETSIZE; k++)
    {
        const int c = (cpu + k) % CPU_SETSIZE;
        if (!CPU_ISSET(c, &allowed) || !CPU_ISSET(c, &l3)) continue;
        if (!CPU_ISSET(c, &smt)) return c;
        if (fallback < 0) fallback = c;
    }
    return fallback;
}
} // namespace

void Resolver1090::helper_main()
{
    if (helper_cpu_ >= 0)
    {
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(helper_cpu_, &one);
        (void)sched_setaffinity(0, sizeof one, &one); // best effort: unpinned it still works, only slower across cache domains
    }
    for (;;)
    {
        Job job;
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return quit_ || job_posted_; });
            if (quit_) return;
            job         = job_;
            job_posted_ = false;
        }
        Walk w;
        w.wall = rate_hz_ ? 0 : std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        uint64_t produced = produced_.load(std::memory_order_relaxed);
        do
        {
            for (unsigned spins = 0; produced - consumed_.load(std::memory_order_acquire) >= kRing; spins++)
            {
                if (spins < 4096) _mm_pause();
                else std::this_thread::yield(); // the consumer is not running (fewer CPUs than threads right now): let it
            }
            Block& blk = blocks_[produced % kRing];
            gate_dispatch(blk, w, job);
            produced_.store(++produced, std::memory_order_release);
        } while (w.i < job.n);
        gate_done_.store(true, std::memory_order_release);
    }
}

// Tried and dropped (round 2): handing the per-aircraft updates to worker threads, aircraft index modulo the worker count -- slower
// everywhere it was measured (the updates are ~10 ns each).  The split here is by what a pass touches instead: the sequential pass owns
// the address table and the gate records, the update pass the published aircraft; they meet in blocks of 1024 frames.
void Resolver1090::gate_dispatch(Block& blk, Walk& w, const Job& job)
{
    if (job.pk) gate_pass<2>(blk, w, job);
    else if (job.dec) gate_pass<0>(blk, w, job);
    else gate_pass<1>(blk, w, job);
}

long Resolver1090::feed(const adsb_amd_record_t* rec, const adsb_amd_decoded_t* dec, size_t n, size_t samples_per_buffer, size_t nbuffers,
                        adsb_amd_on_changed_fn cb, void* user)
{
    Job job;
    job.rec                = rec;
    job.dec                = dec;
    job.n                  = n;
    job.samples_per_buffer = samples_per_buffer;
    return run(job, nbuffers, cb, user);
}

long Resolver1090::feed_packed(const adsb_amd_packed_t* packed, size_t n, size_t samples_per_buffer, size_t nbuffers, adsb_amd_on_changed_fn cb, void* user)
{
    Job job;
    job.pk                 = packed;
    job.n                  = n;
    job.samples_per_buffer = samples_per_buffer;
    return run(job, nbuffers, cb, user);
}

long Resolver1090::run(const Job& job, size_t nbuffers, adsb_amd_on_changed_fn cb, void* user)
{
    const size_t n         = job.n;
    long         accepted  = 0;
    // CPUs this process may run on (its affinity mask: a cgroup, taskset or a NUMA binding can leave one where the machine has many): with a
    // single allowed CPU the caller and the helper would spin against each other a time slice at a time, so everything stays on the caller's thread
    static const unsigned cores = []() -> unsigned {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) return (unsigned)CPU_COUNT(&set);
        return std::thread::hardware_concurrency();
    }();
    static const int      threads = std::getenv("ADSB_AMD_RESOLVER_THREADS") ? std::atoi(std::getenv("ADSB_AMD_RESOLVER_THREADS")) : 2;
    if (n >= kParallelMin && cores > 1 && threads > 1)
    {
        if (!helper_.joinable())
        {
            const char* pin = std::getenv("ADSB_AMD_RESOLVER_PIN"); // 0: leave the helper where the scheduler puts it
            helper_cpu_     = (pin && pin[0] == '0') ? -1 : cache_neighbour(sched_getcpu());
            helper_         = std::thread(&Resolver1090::helper_main, this);
        }
        gate_done_.store(false, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_        = job;
            job_posted_ = true;
        }
        cv_.notify_one();
        uint64_t consumed = consumed_.load(std::memory_order_relaxed);
        for (;;)
        {
            uint64_t produced;
            for (unsigned spins = 0; (produced = produced_.load(std::memory_order_acquire)) == consumed; spins++)
            {
                if (gate_done_.load(std::memory_order_acquire) && produced_.load(std::memory_order_acquire) == consumed) goto drained;
                if (spins < 4096) _mm_pause();
                else std::this_thread::yield(); // the helper is not running: let it
            }
            Block& blk = blocks_[consumed % kRing];
            update_pass(blk, job, cb, user);
            accepted += (long)blk.ne;
            consumed_.store(++consumed, std::memory_order_release);
        }
    drained:;
    }
    else
    {
        Walk w;
        w.wall = rate_hz_ ? 0 : std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        Block& blk = blocks_[0];
        while (w.i < n)
        {
            gate_dispatch(blk, w, job);
            update_pass(blk, job, cb, user);
        }
        accepted = w.accepted;
    }
    stream_base_ += static_cast<uint64_t>(job.samples_per_buffer) * nbuffers;
    return accepted;
}

} // namespace adsb_amd
