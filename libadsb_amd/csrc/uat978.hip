// uat978.hip -- gfx950 kernels for the UAT 978 path: u8 IQ -> phase (reference LUT) -> sign of the phase difference ->
// 18-bit sync search on both sample alignments -> per-candidate sync re-check and frame slicing.
//
// What comes from the reference tree: the phase LUT (UAT978.cpp:76-100) and the map phi = lut[I | Q << 8] (:52).
// Everything after that restates the published dump978 legacy demodulator (un-vendored in the reference):
// parity unpinned, GPU == oracle/oracle978.c is what the tests check.
//
// Round-1 shape (correct first, not yet tuned): three small kernels over a device-resident stream.
//   K1 uat_sign_kernel     one sample per lane: two LUT gathers, wrapped 16-bit difference, __ballot -> 1 bit per sample
//   K2 uat_match_kernel    one 32-sample word per lane, bit-parallel: the 18 check bits sit 2 samples apart, so a match at
//                          sample i is AND_k (S >> 2k) == pattern bit k; 18 funnel shifts shared by both sync words
//   K3 uat_demod_kernel    one wave per candidate (the host may also ask for further sample indices, see uat978_host.cpp): 36-bit sync re-check against the data-derived centre for the candidate
//                          and for the next sample (the reference tries both), then the frame bits sliced at that centre
// The sequential part (which candidate the scan loop reaches, Reed-Solomon, frame choice, skip-ahead) runs on the host.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "uat978.h"

namespace adsb_amd
{
namespace
{
constexpr uint64_t kAdsbSync   = 0xEACDDA4E2ull;
constexpr uint64_t kUplinkSync = 0x153225B1Dull;

__device__ __forceinline__ int phi_difference(uint32_t from, uint32_t to)
{
    // wrap (to - from) into [-32768, 32767]: exactly the int16 reinterpretation of the 16-bit difference
    return (int)(int16_t)(uint16_t)(to - from);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_zero_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ int wave_sum_i(int x)
{
    x += dpp_or_zero_i<0x111, 0xF>(x);
    x += dpp_or_zero_i<0x112, 0xF>(x);
    x += dpp_or_zero_i<0x114, 0xF>(x);
    x += dpp_or_zero_i<0x118, 0xF>(x);
    x += dpp_or_zero_i<0x142, 0xA>(x);
    x += dpp_or_zero_i<0x143, 0xC>(x);
    return __builtin_amdgcn_readlane(x, 63);
}

// ---- K1: sign bit of phi[t+1] - phi[t] for every sample t < n-1 (bit t of the stream, little-endian in 64-bit words)
template <bool PHASES_GIVEN>
__global__ __launch_bounds__(256) void uat_sign_kernel(const uint16_t* __restrict__ in, const uint16_t* __restrict__ lut, uint64_t n,
                                                       uint64_t* __restrict__ signs)
{
    const uint64_t nwords = (n + 63) / 64;
    const uint64_t wave   = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int      lane   = threadIdx.x & 63;
    for (uint64_t w = wave; w < nwords; w += nwaves)
    {
        const uint64_t t = w * 64 + (uint64_t)lane;
        bool           pos = false;
        if (t + 1 < n)
        {
            const uint32_t a = PHASES_GIVEN ? in[t] : lut[in[t]];
            const uint32_t b = PHASES_GIVEN ? in[t + 1] : lut[in[t + 1]];
            pos              = phi_difference(a, b) > 0;
        }
        const uint64_t m = __ballot(pos);
        if (lane == 0) signs[w] = m;
    }
}

// ---- K2: sample indices i whose 18 stride-2 sign bits equal the top 18 bits of a sync word
__global__ __launch_bounds__(256) void uat_match_kernel(const uint64_t* __restrict__ signs, uint64_t n, uint32_t* __restrict__ cand,
                                                        uint32_t cap, uint32_t* __restrict__ count)
{
    // bit k of the window (k = 0 first / oldest) must equal bit (35 - k) of the 36-bit sync word, k = 0..17
    const uint64_t nword32 = (n + 31) / 32;
    const uint64_t tid     = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride  = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t* s32    = reinterpret_cast<const uint32_t*>(signs);
    const uint64_t navail  = ((n + 63) / 64) * 2; // 32-bit words backed by memory
    for (uint64_t w = tid; w < nword32; w += stride)
    {
        const uint32_t w0 = s32[w];
        const uint32_t w1 = (w + 1 < navail) ? s32[w + 1] : 0u;
        const uint32_t w2 = (w + 2 < navail) ? s32[w + 2] : 0u;
        uint32_t accA = 0xFFFFFFFFu, accU = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 18; k++)
        {
            const int sh = 2 * k; // 0..34
            uint32_t  v;          // bit j = sign bit of sample 32 w + j + 2k
            if (sh == 0) v = w0;
            else if (sh < 32) v = __builtin_amdgcn_alignbit(w1, w0, sh);
            else if (sh == 32) v = w1;
            else v = __builtin_amdgcn_alignbit(w2, w1, sh - 32);
            accA &= ((kAdsbSync >> (35 - k)) & 1ull) ? v : ~v;
            accU &= ((kUplinkSync >> (35 - k)) & 1ull) ? v : ~v;
        }
        // the window of a match at sample i ends at i + 34 and the difference needs sample i + 35
        uint32_t hits = accA | accU;
        while (hits)
        {
            const int      j = __builtin_ctz(hits);
            hits &= hits - 1;
            const uint64_t i = w * 32 + (uint64_t)j;
            if (i + 36 > n) continue;
            const uint32_t kind = ((accA >> j) & 1u) ? 0u : 1u; // both cannot match (different first bit)
            const uint32_t slot = atomicAdd(count, 1u);
            if (slot < cap) cand[slot] = ((uint32_t)i & 0x7FFFFFFFu) | (kind << 31);
        }
    }
}

// ---- K3: one wave per candidate
// 64 sign bits starting at sample p (bit 0 = sample p); 0 beyond the stream.  `signs` has two zeroed words of slack.
__device__ __forceinline__ uint64_t sign_window(const uint64_t* __restrict__ signs, uint64_t n, uint64_t p)
{
    if (p >= n) return 0;
    const uint64_t w = p >> 6;
    const int      r = (int)(p & 63);
    uint64_t       v = signs[w] >> r;
    if (r) v |= signs[w + 1] << (64 - r);
    return v;
}
struct SyncCheck
{
    bool ok;
    int  center;
};

// check_sync_word: centre = mean of the per-class means of dphi over the 36 sync bits (C integer division), then at most
// four bits on the wrong side of it
template <bool PHASES_GIVEN>
__device__ __forceinline__ int dphi_at(const uint16_t* __restrict__ in, const uint16_t* __restrict__ lut, uint64_t n, uint64_t s)
{
    if (s + 1 >= n) return 0;
    const uint32_t a = PHASES_GIVEN ? in[s] : lut[in[s]];
    const uint32_t b = PHASES_GIVEN ? in[s + 1] : lut[in[s + 1]];
    return phi_difference(a, b);
}

template <bool PHASES_GIVEN>
__device__ __forceinline__ SyncCheck check_sync(const uint16_t* __restrict__ in, const uint16_t* __restrict__ lut, uint64_t n, uint64_t start,
                                                uint64_t pattern, int lane)
{
    const bool in_sync = lane < 36;
    const int  d       = in_sync ? dphi_at<PHASES_GIVEN>(in, lut, n, start + 2ull * (uint64_t)lane) : 0;
    const bool one     = in_sync && ((pattern >> ((35 - lane) & 63)) & 1ull);
    const bool zero    = in_sync && !one;
    const int  ones    = __builtin_popcountll(__ballot(one)), zeros = __builtin_popcountll(__ballot(zero));
    const int  one_tot = wave_sum_i(one ? d : 0), zero_tot = wave_sum_i(zero ? d : 0);
    SyncCheck  r;
    r.center      = (int)(int16_t)((one_tot / ones + zero_tot / zeros) / 2);
    const bool bad = (one && d < r.center) || (zero && d > r.center);
    r.ok          = __builtin_popcountll(__ballot(bad)) <= 4;
    return r;
}

// slice `nbits` frame bits starting at sample `start` (first bit after the sync word), MSB-first bytes into out[]
template <bool PHASES_GIVEN>
__device__ __forceinline__ void slice_frame(const uint16_t* __restrict__ in, const uint16_t* __restrict__ lut, uint64_t n, uint64_t start,
                                            int center, int nbits, uint8_t* __restrict__ out, int lane)
{
    for (int base = 0; base < nbits; base += 64)
    {
        const int      b    = base + lane;
        const int      d    = (b < nbits) ? dphi_at<PHASES_GIVEN>(in, lut, n, start + 2ull * (uint64_t)b) : 0;
        const uint64_t bits = __ballot(b < nbits && d > center); // bit `lane` = frame bit base + lane
        if (lane < 8 && base + 8 * lane < nbits)
        { // byte k of this group = frame bits base + 8k .. 8k + 7, first bit = MSB
            const uint32_t byte = (uint32_t)(bits >> (8 * lane)) & 0xFFu;
            out[(base >> 3) + lane] = (uint8_t)(__builtin_bitreverse32(byte) >> 24);
        }
    }
}

template <bool PHASES_GIVEN>
__global__ __launch_bounds__(64) void uat_demod_kernel(const uint16_t* __restrict__ in, const uint16_t* __restrict__ lut, uint64_t n,
                                                       const uint64_t* __restrict__ signs, const uint32_t* __restrict__ cand, uint32_t ncand, uat_adsb_rec_t* __restrict__ adsb,
                                                       uat_uplink_rec_t* __restrict__ uplink, uint32_t uplink_cap,
                                                       uint32_t* __restrict__ uplink_count)
{
    const int lane = threadIdx.x;
    for (uint32_t c = blockIdx.x; c < ncand; c += gridDim.x)
    {
        const uint32_t raw  = cand[c];
        const uint32_t kind = raw >> 31;
        const uint64_t idx  = raw & 0x7FFFFFFFu;
        if (kind == 0)
        {
            uat_adsb_rec_t* r = &adsb[c];
            if (lane == 0)
            {
                const uint64_t sb = idx >> 1;
                r->index    = (uint32_t)idx;
                r->kind     = 0;
                r->window   = sign_window(signs, n, 2 * sb);
                r->after[0] = sign_window(signs, n, 2 * (sb + 36 + 240 + 1));
                r->after[1] = sign_window(signs, n, 2 * (sb + 36 + 384 + 1));
            }
#pragma unroll
            for (int v = 0; v < 2; v++)
            {
                const SyncCheck sc = check_sync<PHASES_GIVEN>(in, lut, n, idx + (uint64_t)v, kAdsbSync, lane);
                if (lane == 0)
                {
                    r->ok[v]     = sc.ok ? 1 : 0;
                    r->center[v] = (int16_t)sc.center;
                }
                if (sc.ok) slice_frame<PHASES_GIVEN>(in, lut, n, idx + (uint64_t)v + 72, sc.center, 384, r->frame[v], lane);
            }
        }
        else
        {
            if (lane == 0)
            {
                const uint64_t sb = idx >> 1;
                adsb[c].index    = (uint32_t)idx;
                adsb[c].kind     = 1;
                adsb[c].window   = sign_window(signs, n, 2 * sb);
                adsb[c].after[0] = sign_window(signs, n, 2 * (sb + 36 + 4416 + 1));
                adsb[c].after[1] = 0;
            }
            uint32_t slot = 0;
            if (lane == 0) slot = atomicAdd(uplink_count, 1u);
            slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
            if (lane == 0) adsb[c].uplink_slot = slot;
            if (slot >= uplink_cap) continue;
            uat_uplink_rec_t* r = &uplink[slot];
#pragma unroll 1
            for (int v = 0; v < 2; v++)
            {
                const SyncCheck sc = check_sync<PHASES_GIVEN>(in, lut, n, idx + (uint64_t)v, kUplinkSync, lane);
                if (lane == 0)
                {
                    r->ok[v]     = sc.ok ? 1 : 0;
                    r->center[v] = (int16_t)sc.center;
                }
                if (sc.ok) slice_frame<PHASES_GIVEN>(in, lut, n, idx + (uint64_t)v + 72, sc.center, 4416, r->frame[v], lane);
            }
        }
    }
}
} // namespace

hipError_t launch_uat978(const UatArgs& a, hipStream_t stream)
{
    if (a.nsamples < 2) return hipMemsetAsync(a.counts, 0, 2 * sizeof(uint32_t), stream);
    hipError_t e = hipMemsetAsync(a.counts, 0, 2 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const uint64_t nwords = (a.nsamples + 63) / 64;
    e = hipMemsetAsync(a.signs + nwords, 0, 2 * sizeof(uint64_t), stream);
    if (e != hipSuccess) return e;
    uint32_t       g1     = (uint32_t)((nwords + 3) / 4 > 8192 ? 8192 : (nwords + 3) / 4);
    if (a.phases_given) hipLaunchKernelGGL(uat_sign_kernel<true>, dim3(g1), dim3(256), 0, stream, a.in, a.lut, a.nsamples, a.signs);
    else hipLaunchKernelGGL(uat_sign_kernel<false>, dim3(g1), dim3(256), 0, stream, a.in, a.lut, a.nsamples, a.signs);
    const uint64_t nw32 = (a.nsamples + 31) / 32;
    uint32_t       g2   = (uint32_t)((nw32 + 255) / 256 > 4096 ? 4096 : (nw32 + 255) / 256);
    hipLaunchKernelGGL(uat_match_kernel, dim3(g2), dim3(256), 0, stream, a.signs, a.nsamples, a.cand, a.cand_cap, a.counts);
    return hipGetLastError();
}

hipError_t launch_uat978_demod(const UatArgs& a, uint32_t ncand, hipStream_t stream)
{
    if (ncand == 0) return hipSuccess;
    uint32_t g = ncand > 4096 ? 4096 : ncand;
    if (a.phases_given)
        hipLaunchKernelGGL(uat_demod_kernel<true>, dim3(g), dim3(64), 0, stream, a.in, a.lut, a.nsamples, a.signs, a.cand, ncand, a.adsb, a.uplink, a.uplink_cap,
                           a.counts + 1);
    else
        hipLaunchKernelGGL(uat_demod_kernel<false>, dim3(g), dim3(64), 0, stream, a.in, a.lut, a.nsamples, a.signs, a.cand, ncand, a.adsb, a.uplink, a.uplink_cap,
                           a.counts + 1);
    return hipGetLastError();
}

} // namespace adsb_amd
