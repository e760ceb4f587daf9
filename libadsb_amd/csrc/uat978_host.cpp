// uat978_host.cpp -- host half of the UAT 978 path and its C ABI (include/adsb_amd.h, "UAT 978" section).
//
//   GPU (uat978.hip)   sign of the phase difference for every sample, every exact 18-bit sync match, and for each match
//                      the 36-bit sync re-check + sliced frame for the match and for the next sample
//   host (this file)   what the dump978 scan loop does with those: which match the loop reaches (it jumps over a decoded
//                      frame and does not clear its two shift registers when it does), Reed-Solomon, choice between the
//                      two slicings, up-call.  Plus UAT978Handler::HandleData's staging/re-buffering (UAT978.cpp:43-60).
//
// The algorithm restated here is the published dump978 legacy demodulator; it is un-vendored in the reference tree
// (SURVEY.md F7), so parity is unpinned: tests compare this path with oracle/oracle978.c on generated streams.
// There is no CPU demodulator in this file: phases, signs, matches and slicing only ever come from the device.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "adsb_amd.h"
#include "scan1090.h"
#include "uat978.h"

using namespace adsb_amd;

namespace
{
// ------------------------------------------------------------------------------------------------------------------
// Reed-Solomon over GF(256), field polynomial 0x187, first consecutive root 120, primitive element 1: the three codes
// init_fec() sets up (RS(30,18), RS(48,34), RS(92,72) as shortened RS(255, 255 - nroots)).  The decoder follows the
// classic Berlekamp-Massey / Chien / Forney procedure with libfec's conventions, because results on words that are
// NOT within the correction radius (or whose "errors" fall into the zero padding) depend on the procedure:
// corrections located in the padding are dropped but still counted, and a zero Forney denominator is not rejected.
// ------------------------------------------------------------------------------------------------------------------
class ReedSolomon
{
  public:
    ReedSolomon(int nroots, int pad) : nroots_(nroots), pad_(pad)
    {
        int x = 1;
        for (int i = 0; i < 255; i++)
        {
            exp_[i] = exp_[i + 255] = (uint8_t)x;
            log_[x]                 = i;
            x <<= 1;
            if (x & 0x100) x ^= 0x187;
        }
        log_[0] = -1;
    }
    int codeword_bytes() const { return 255 - pad_; }

    // in place; returns the number of located errors or -1 (data untouched)
    int decode(uint8_t* data) const
    {
        const int nr = nroots_, n = 255 - pad_;
        uint8_t   s[32];
        bool      any = false;
        for (int i = 0; i < nr; i++)
        {
            uint8_t   acc = 0;
            const int a   = (kFcr + i) % 255;
            for (int j = 0; j < n; j++) acc = (uint8_t)(mul_exp(acc, a) ^ data[j]);
            s[i] = acc;
            any |= acc != 0;
        }
        if (!any) return 0;

        uint8_t lambda[33] = {1}, b[33] = {1}, t[33];
        int     el = 0;
        for (int r = 1; r <= nr; r++)
        {
            uint8_t discr = 0;
            for (int i = 0; i < r; i++) discr ^= mul(lambda[i], s[r - i - 1]);
            if (discr == 0)
            {
                std::memmove(b + 1, b, (size_t)nr);
                b[0] = 0;
                continue;
            }
            t[0] = lambda[0];
            for (int i = 0; i < nr; i++) t[i + 1] = (uint8_t)(lambda[i + 1] ^ mul(discr, b[i]));
            if (2 * el <= r - 1)
            {
                el = r - el;
                for (int i = 0; i <= nr; i++) b[i] = div(lambda[i], discr);
            }
            else
            {
                std::memmove(b + 1, b, (size_t)nr);
                b[0] = 0;
            }
            std::memcpy(lambda, t, (size_t)nr + 1);
        }
        int deg = 0;
        for (int i = 0; i <= nr; i++)
            if (lambda[i]) deg = i;

        // roots of lambda: X^-1 = alpha^i  <=>  error at position (i - 1) counted from the end of the full 255 word
        int root[32], loc[32], count = 0;
        for (int i = 1; i <= 255 && count < deg; i++)
        {
            uint8_t q = 1;
            for (int j = 1; j <= deg; j++)
                if (lambda[j]) q ^= exp_[(log_[lambda[j]] + j * i) % 255];
            if (q) continue;
            root[count] = i;
            loc[count]  = i - 1;
            count++;
        }
        if (count != deg) return -1;

        uint8_t omega[33];
        for (int i = 0; i < deg; i++)
        {
            uint8_t acc = 0;
            for (int j = 0; j <= i; j++) acc ^= mul(s[i - j], lambda[j]);
            omega[i] = acc;
        }
        for (int j = count - 1; j >= 0; j--)
        {
            uint8_t num1 = 0;
            for (int i = deg - 1; i >= 0; i--)
                if (omega[i]) num1 ^= exp_[(log_[omega[i]] + i * root[j]) % 255];
            const uint8_t num2 = exp_[(root[j] * (kFcr - 1) + 255) % 255];
            uint8_t       den  = 0;
            for (int i = std::min(deg, nr - 1) & ~1; i >= 0; i -= 2)
                if (lambda[i + 1]) den ^= exp_[(log_[lambda[i + 1]] + i * root[j]) % 255];
            if (num1 != 0 && loc[j] >= pad_)
            {
                const int lden = den ? log_[den] : 255; // index form of zero is 255 in libfec: the exponent then gains 255 - 255
                data[loc[j] - pad_] ^= exp_[(log_[num1] + log_[num2] + 255 - lden) % 255];
            }
        }
        return count;
    }

  private:
    static constexpr int kFcr = 120;
    uint8_t              mul(uint8_t a, uint8_t b) const { return (a && b) ? exp_[log_[a] + log_[b]] : 0; }
    uint8_t              mul_exp(uint8_t a, int e) const { return a ? exp_[log_[a] + e] : 0; }
    uint8_t              div(uint8_t a, uint8_t b) const { return a ? exp_[log_[a] + 255 - log_[b]] : 0; }
    int                  nroots_, pad_;
    uint8_t              exp_[510];
    int                  log_[256];
};

struct Fec
{
    ReedSolomon adsb_short{12, 225}, adsb_long{14, 207}, uplink{20, 163};
};
const Fec& fec()
{
    static const Fec f;
    return f;
}

constexpr int      kShortSkip = 36 + 240, kLongSkip = 36 + 384, kUplinkSkip = 36 + kUatUplinkBits;
constexpr uint32_t kCheckMask = (1u << kUatCheckBits) - 1u;
constexpr uint32_t kCheckAdsb = (uint32_t)(0xEACDDA4E2ull >> 18), kCheckUplink = (uint32_t)(0x153225B1Dull >> 18);

// correct_adsb_frame: long first, in place; then short on whatever the long attempt left.  skip in bits, 0 = neither.
int correct_adsb(uint8_t* f, int* rs)
{
    int n = fec().adsb_long.decode(f);
    if (n >= 0 && n <= 7 && (f[0] >> 3) != 0)
    {
        *rs = n;
        return kLongSkip;
    }
    n = fec().adsb_short.decode(f);
    if (n >= 0 && n <= 6 && (f[0] >> 3) == 0)
    {
        *rs = n;
        return kShortSkip;
    }
    *rs = 9999;
    return 0;
}

int correct_uplink(const uint8_t* raw, uint8_t* out, int* rs)
{
    int total = 0;
    for (int block = 0; block < 6; block++)
    {
        uint8_t cw[92];
        for (int i = 0; i < 92; i++) cw[i] = raw[i * 6 + block];
        const int n = fec().uplink.decode(cw);
        if (n < 0 || n > 10)
        {
            *rs = 9999;
            return 0;
        }
        total += n;
        std::memcpy(out + block * 72, cw, 72);
    }
    *rs = total;
    return kUplinkSkip;
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
} // namespace

struct adsb_amd_uat
{
    int         device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t  ev[4]  = {nullptr, nullptr, nullptr, nullptr};
    std::string error;

    uint16_t* lut_d = nullptr;
    std::vector<uint16_t> lut_h;

    // scratch sized to the largest stream seen
    uint64_t*         signs_d = nullptr;
    size_t            signs_words = 0;
    uint32_t*         cand_d = nullptr;
    uint32_t          cand_cap = 0;
    uint32_t*         counts_d = nullptr;
    uint32_t*         counts_h = nullptr; // pinned
    uat_adsb_rec_t*   adsb_d = nullptr;
    uat_uplink_rec_t* uplink_d = nullptr;
    uint32_t          uplink_cap = 0;
    std::vector<uat_adsb_rec_t>   adsb_h;
    std::vector<uat_uplink_rec_t> uplink_h;
    std::unordered_map<uint64_t, uint32_t> by_key; // (index << 1 | kind) -> record
    uint32_t nrecords = 0, nuplink = 0;

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
    uint64_t stat_candidates = 0, stat_extra = 0;

    ~adsb_amd_uat()
    {
        (void)hipSetDevice(device);
        for (void* p : {(void*)lut_d, (void*)signs_d, (void*)cand_d, (void*)counts_d, (void*)adsb_d, (void*)uplink_d, (void*)in_d, (void*)stage_d,
                        (void*)stage_tmp_d})
            if (p) (void)hipFree(p);
        if (counts_h) (void)hipHostFree(counts_h);
        for (auto e : ev)
            if (e) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
    }

    int init()
    {
        UAT_HIP(hipSetDevice(device));
        UAT_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (auto& e : ev) UAT_HIP(hipEventCreate(&e));
        // InitATan2Table, UAT978.cpp:76-100
        lut_h.resize(65536);
        for (unsigned i = 0; i < 256; i++)
            for (unsigned q = 0; q < 256; q++)
            {
                const double ang    = std::atan2((double)q - 127.5, (double)i - 127.5) + M_PI;
                const double scaled = std::round(32768 * ang / M_PI);
                lut_h[i | (q << 8)] = (uint16_t)(scaled < 0 ? 0 : scaled > 65535 ? 65535 : scaled);
            }
        UAT_HIP(hipMalloc(&lut_d, 65536 * sizeof(uint16_t)));
        UAT_HIP(hipMemcpy(lut_d, lut_h.data(), 65536 * sizeof(uint16_t), hipMemcpyHostToDevice));
        UAT_HIP(hipMalloc(&counts_d, 2 * sizeof(uint32_t)));
        UAT_HIP(hipHostMalloc(&counts_h, 2 * sizeof(uint32_t)));
        UAT_HIP(hipMalloc(&stage_d, 65536 * sizeof(uint16_t)));
        UAT_HIP(hipMalloc(&stage_tmp_d, 65536 * sizeof(uint16_t)));
        UAT_HIP(hipMemset(stage_d, 0, 65536 * sizeof(uint16_t)));
        return ADSB_AMD_OK;
    }

    int reserve(uint64_t nsamples, uint32_t want_cand, uint32_t want_uplink)
    {
        const size_t words = (size_t)((nsamples + 63) / 64) + 2;
        if (words > signs_words)
        {
            if (signs_d) (void)hipFree(signs_d);
            signs_d = nullptr, signs_words = 0;
            UAT_HIP(hipMalloc(&signs_d, words * sizeof(uint64_t)));
            signs_words = words;
        }
        if (want_cand > cand_cap)
        {
            if (cand_d) (void)hipFree(cand_d);
            if (adsb_d) (void)hipFree(adsb_d);
            cand_d = nullptr, adsb_d = nullptr, cand_cap = 0;
            UAT_HIP(hipMalloc(&cand_d, (size_t)want_cand * sizeof(uint32_t)));
            UAT_HIP(hipMalloc(&adsb_d, (size_t)want_cand * sizeof(uat_adsb_rec_t)));
            cand_cap = want_cand;
        }
        if (want_uplink > uplink_cap)
        {
            if (uplink_d) (void)hipFree(uplink_d);
            uplink_d = nullptr, uplink_cap = 0;
            UAT_HIP(hipMalloc(&uplink_d, (size_t)want_uplink * sizeof(uat_uplink_rec_t)));
            uplink_cap = want_uplink;
        }
        return ADSB_AMD_OK;
    }

    UatArgs args(const uint16_t* in, uint64_t n, bool phases_given) const
    {
        UatArgs a{};
        a.in = in, a.lut = lut_d, a.nsamples = n, a.phases_given = phases_given ? 1 : 0;
        a.signs = signs_d, a.cand = cand_d, a.cand_cap = cand_cap, a.counts = counts_d;
        a.adsb = adsb_d, a.uplink = uplink_d, a.uplink_cap = uplink_cap;
        return a;
    }

    // GPU part of one process_buffer: after this adsb_h / uplink_h hold one record per exact 18-bit match
    int scan(const uint16_t* in_dev, uint64_t n, bool phases_given)
    {
        if (n >= (1ull << 31)) return fail(ADSB_AMD_EINVAL, "UAT stream longer than 2^31 samples per call");
        int rc = reserve(phases_given ? n : 0, std::max<uint32_t>(cand_cap, 4096), std::max<uint32_t>(uplink_cap, 1024)); // sign words: phases path only
        if (rc) return rc;
        nrecords = nuplink = 0;
        by_key.clear();
        for (int attempt = 0;; attempt++)
        {
            const UatArgs a = args(in_dev, n, phases_given);
            UAT_HIP(hipEventRecord(ev[0], stream));
            UAT_HIP(launch_uat978(a, stream));
            UAT_HIP(hipEventRecord(ev[1], stream));
            UAT_HIP(hipMemcpyAsync(counts_h, counts_d, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            UAT_HIP(hipStreamSynchronize(stream));
            const uint32_t ncand = counts_h[0];
            if (ncand > cand_cap)
            { // the match kernel counted past the array: grow and repeat the match
                if (attempt > 2) return fail(ADSB_AMD_EHIP, "UAT candidate count keeps changing");
                rc = reserve(0, ncand + ncand / 4 + 64, uplink_cap);
                if (rc) return rc;
                continue;
            }
            if (ncand > uplink_cap)
            {
                rc = reserve(0, cand_cap, ncand + 64);
                if (rc) return rc;
            }
            stat_candidates += ncand;
            rc = demod_on_device(in_dev, n, phases_given, ncand, 0);
            if (rc) return rc;
            UAT_HIP(hipEventElapsedTime(&scan_ms, ev[0], ev[1]));
            return ADSB_AMD_OK;
        }
    }

    // run K3 over cand_d[first .. first + count) and append the records to adsb_h / uplink_h
    int demod_on_device(const uint16_t* in_dev, uint64_t n, bool phases_given, uint32_t count, uint32_t first)
    {
        if (count == 0) return ADSB_AMD_OK;
        UatArgs a = args(in_dev, n, phases_given);
        a.cand += first;
        a.adsb += first;
        UAT_HIP(hipEventRecord(ev[2], stream));
        UAT_HIP(launch_uat978_demod(a, count, stream));
        UAT_HIP(hipEventRecord(ev[3], stream));
        UAT_HIP(hipMemcpyAsync(counts_h + 1, counts_d + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        adsb_h.resize((size_t)first + count);
        UAT_HIP(hipMemcpyAsync(adsb_h.data() + first, adsb_d + first, (size_t)count * sizeof(uat_adsb_rec_t), hipMemcpyDeviceToHost, stream));
        UAT_HIP(hipStreamSynchronize(stream));
        const uint32_t up_total = counts_h[1];
        if (up_total > uplink_cap) return fail(ADSB_AMD_EHIP, "UAT uplink record area too small");
        if (up_total > nuplink)
        {
            uplink_h.resize(up_total);
            UAT_HIP(hipMemcpy(uplink_h.data() + nuplink, uplink_d + nuplink, (size_t)(up_total - nuplink) * sizeof(uat_uplink_rec_t),
                              hipMemcpyDeviceToHost));
            nuplink = up_total;
        }
        for (uint32_t k = first; k < first + count; k++) by_key[((uint64_t)adsb_h[k].index << 1) | adsb_h[k].kind] = k;
        nrecords = first + count;
        float ms = 0.f;
        UAT_HIP(hipEventElapsedTime(&ms, ev[2], ev[3]));
        demod_ms = first ? demod_ms + ms : ms;
        return ADSB_AMD_OK;
    }

    // record for (index, kind); asks the device when the scan loop reaches a position that is not an exact match in the
    // stream (possible only through stale register bits right after a jump)
    int record_for(const uint16_t* in_dev, uint64_t n, bool phases_given, uint32_t index, uint32_t kind, uint32_t* out)
    {
        auto it = by_key.find(((uint64_t)index << 1) | kind);
        if (it != by_key.end())
        {
            *out = it->second;
            return ADSB_AMD_OK;
        }
        if (nrecords + 1 > cand_cap || nuplink + 1 > uplink_cap)
        { // grow, keeping what is there (rare path; host copies are authoritative, the device arrays only receive)
            int rc = grow_keep();
            if (rc) return rc;
        }
        const uint32_t raw = (index & 0x7FFFFFFFu) | (kind << 31);
        UAT_HIP(hipMemcpyAsync(cand_d + nrecords, &raw, sizeof(raw), hipMemcpyHostToDevice, stream));
        UAT_HIP(hipStreamSynchronize(stream));
        stat_extra++;
        const uint32_t at = nrecords;
        int            rc = demod_on_device(in_dev, n, phases_given, 1, at);
        if (rc) return rc;
        *out = at;
        return ADSB_AMD_OK;
    }

    int grow_keep()
    {
        // new arrays twice the size; device contents are not needed again (records already copied to the host), but the
        // uplink counter keeps counting from nuplink so new records land behind the old ones
        uint32_t* oc = cand_d;
        auto*     oa = adsb_d;
        auto*     ou = uplink_d;
        cand_d = nullptr, adsb_d = nullptr, uplink_d = nullptr;
        const uint32_t nc = cand_cap * 2 + 64, nu = uplink_cap * 2 + 64;
        cand_cap = uplink_cap = 0;
        int rc = reserve(0, nc, nu);
        (void)hipFree(oc), (void)hipFree(oa), (void)hipFree(ou);
        return rc;
    }

    int fail(int code, const char* msg)
    {
        error = msg;
        return code;
    }

    // ---------------------------------------------------------------------------------------------------------------
    // process_buffer: the dump978 scan loop over the device's matches.  `len` samples, returns samples consumed.
    // ---------------------------------------------------------------------------------------------------------------
    struct Attempt
    {
        int     skip = 0, rs = 9999, len = 0, variant = 0;
        uint8_t data[432];
    };

    // demod_*_frame at index and index + 1, then the reference's choice between them
    bool attempt(const uat_adsb_rec_t& r, Attempt& best)
    {
        int     skip[2] = {0, 0}, rs[2] = {-1, -1};
        uint8_t buf[2][432];
        for (int v = 0; v < 2; v++)
        {
            if (r.kind == 0)
            {
                if (!r.ok[v])
                {
                    rs[v] = 9999;
                    continue;
                }
                std::memcpy(buf[v], r.frame[v], kUatLongBytes);
                skip[v] = correct_adsb(buf[v], &rs[v]);
            }
            else
            {
                const uat_uplink_rec_t& u = uplink_h[r.uplink_slot];
                if (!u.ok[v])
                {
                    rs[v] = 9999;
                    continue;
                }
                skip[v] = correct_uplink(u.frame[v], buf[v], &rs[v]);
            }
        }
        int v;
        if (skip[0] && rs[0] <= rs[1]) v = 0;
        else if (skip[1] && rs[1] <= rs[0]) v = 1;
        else return false;
        best.skip = skip[v], best.rs = rs[v], best.variant = v;
        best.len = r.kind ? 432 : ((buf[v][0] >> 3) == 0 ? 18 : 34);
        std::memcpy(best.data, buf[v], (size_t)best.len);
        return true;
    }

    static uint32_t reg_from_window(uint64_t window, int alignment)
    { // 18 sign bits two samples apart, oldest first = most significant
        uint32_t r = 0;
        for (int k = 0; k < 18; k++) r = (r << 1) | (uint32_t)((window >> (2 * k + alignment)) & 1u);
        return r;
    }

    int process(const uint16_t* in_dev, uint64_t len, bool phases_given, uint64_t stream_offset, adsb_amd_uat_frame_fn cb, void* user,
                int64_t* consumed)
    {
        int rc = scan(in_dev, len, phases_given);
        if (rc) return rc;
        const int64_t lenbits = (int64_t)(len / 2) - (kUatSyncBits + kUatUplinkBits);

        // exact matches in stream order, grouped by start bit
        std::vector<uint32_t> order(nrecords);
        for (uint32_t k = 0; k < nrecords; k++) order[k] = k;
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return adsb_h[x].index < adsb_h[y].index; });

        int64_t bit = 0; // next bit the loop will examine
        size_t  pos = 0;
        auto emit = [&](const uat_adsb_rec_t& r, const Attempt& a)
        {
            if (cb) cb(user, r.kind ? '+' : '-', a.data, a.len, a.rs, stream_offset + r.index + (uint64_t)a.variant);
        };

        while (bit < lenbits)
        {
            // --- registers hold only stream bits: the loop fires exactly at the device's matches
            while (pos < order.size() && (int64_t)(adsb_h[order[pos]].index >> 1) + 17 < std::max<int64_t>(bit, kUatCheckBits)) pos++;
            if (pos >= order.size()) break;
            const int64_t startbit = adsb_h[order[pos]].index >> 1;
            if (startbit + 17 >= lenbits) break;
            // matches at this start bit: even/odd sample, ADS-B/uplink word
            const uat_adsb_rec_t* m[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; // [kind][alignment]
            size_t                q       = pos;
            for (; q < order.size() && (int64_t)(adsb_h[order[q]].index >> 1) == startbit; q++)
            {
                const uat_adsb_rec_t& r = adsb_h[order[q]];
                m[r.kind][r.index & 1]  = &r;
            }
            pos = q;
            bit = startbit + 17; // the loop is at this bit now
            const int             kind = (m[0][0] || m[0][1]) ? 0 : 1; // `else if`: the uplink word is only looked at without an ADS-B match
            const uat_adsb_rec_t* r    = m[kind][0] ? m[kind][0] : m[kind][1];
            Attempt               a;
            if (!attempt(*r, a))
            {
                bit++;
                continue;
            }
            emit(*r, a);
            // --- jump: bit = startbit + skip, then the loop's ++.  The registers keep their contents, so for the next 17 bits
            // they mix bits from before the jump with new ones and can fire where the stream itself has no match.
            uint32_t reg[2]  = {reg_from_window(r->window, 0), reg_from_window(r->window, 1)};
            uint64_t fresh   = r->kind ? r->after[0] : (a.skip == kShortSkip ? r->after[0] : r->after[1]);
            bit              = startbit + a.skip + 1;
            int64_t mixed_until = bit + 17; // first bit at which both registers hold 18 new bits again
            int     t        = 0;
            while (bit < lenbits && bit < mixed_until)
            {
                reg[0] = ((reg[0] << 1) | (uint32_t)((fresh >> (2 * t)) & 1u)) & kCheckMask;
                reg[1] = ((reg[1] << 1) | (uint32_t)((fresh >> (2 * t + 1)) & 1u)) & kCheckMask;
                t++;
                uint32_t k2;
                if (reg[0] == kCheckAdsb || reg[1] == kCheckAdsb) k2 = 0;
                else if (reg[0] == kCheckUplink || reg[1] == kCheckUplink) k2 = 1;
                else
                {
                    bit++;
                    continue;
                }
                const uint32_t want  = k2 ? kCheckUplink : kCheckAdsb;
                const int64_t  sb2   = bit - kUatCheckBits + 1;
                const uint32_t index = (uint32_t)(sb2 * 2 + (reg[0] == want ? 0 : 1));
                uint32_t       at    = 0;
                rc                   = record_for(in_dev, len, phases_given, index, k2, &at);
                if (rc) return rc;
                const uat_adsb_rec_t r2 = adsb_h[at]; // copy: adsb_h may grow below
                Attempt              a2;
                if (!attempt(r2, a2))
                {
                    bit++;
                    continue;
                }
                emit(r2, a2);
                fresh       = r2.kind ? r2.after[0] : (a2.skip == kShortSkip ? r2.after[0] : r2.after[1]);
                bit         = sb2 + a2.skip + 1;
                mixed_until = bit + 17;
                t           = 0;
            }
        }
        if (bit < lenbits) bit = lenbits; // no further match: the loop runs to the end
        *consumed = lenbits > 0 ? (bit - kUatCheckBits) * 2 : (int64_t)-2 * kUatCheckBits;
        return ADSB_AMD_OK;
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
extern "C" int adsb_amd_uat_handle_data(adsb_amd_uat_t* u, const uint8_t* iq_host, size_t nbytes, adsb_amd_uat_frame_fn cb, void* user)
{
    if (!u || (!iq_host && nbytes)) return ADSB_AMD_EINVAL;
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
    if (hipSetDevice(u->device) != hipSuccess) return u->fail(ADSB_AMD_EHIP, "hipSetDevice failed");
    int rc = u->upload(phi_host, (size_t)len * 2);
    if (rc) return rc;
    return u->process(reinterpret_cast<const uint16_t*>(u->in_d), len, true, offset, cb, user, consumed);
}
extern "C" int adsb_amd_uat_process_iq(adsb_amd_uat_t* u, const void* iq, uint64_t nsamples, int on_device, uint64_t offset,
                                       adsb_amd_uat_frame_fn cb, void* user, int64_t* consumed)
{
    if (!u || !consumed || (!iq && nsamples)) return ADSB_AMD_EINVAL;
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
extern "C" int adsb_amd_uat_timing(const adsb_amd_uat_t* u, float* scan_ms, float* demod_ms, uint64_t* candidates, uint64_t* extra_lookups)
{
    if (!u) return ADSB_AMD_EINVAL;
    if (scan_ms) *scan_ms = u->scan_ms;
    if (demod_ms) *demod_ms = u->demod_ms;
    if (candidates) *candidates = u->stat_candidates;
    if (extra_lookups) *extra_lookups = u->stat_extra;
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_uat_phase_lut(const adsb_amd_uat_t* u, uint16_t* lut65536)
{
    if (!u || !lut65536) return ADSB_AMD_EINVAL;
    std::memcpy(lut65536, u->lut_h.data(), 65536 * sizeof(uint16_t));
    return ADSB_AMD_OK;
}
extern "C" int adsb_amd_uat_rs_decode(int kind, uint8_t* codeword)
{
    const ReedSolomon& rs = kind == 0 ? fec().adsb_short : kind == 1 ? fec().adsb_long : fec().uplink;
    return rs.decode(codeword);
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
    (void)fec();
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
