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
__global__ __launch_bounds__(256) void uat_sign_kernel(const uint16_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ signs)
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
            pos = phi_difference(in[t], in[t + 1]) > 0;
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

// ---- K1+K2 fused for u8 IQ input (the batch path): the discriminator and the sync search in one pass over HBM.
// One 1024-lane workgroup per CU, persistent.  LDS holds the whole phase LUT (128 KiB; the IQ pair read as one u16 IS
// the index, UAT978.cpp:52) plus one chunk's sign bits.  Per chunk of 32 768 samples:
//   A  each wave takes 4 rows of 512 samples, 16 B (8 samples) per lane per row: 8 LDS gathers give the phases as packed
//      u16 pairs, the pair shifted by one sample comes from v_alignbit, v_pk_sub_i16 is the wrapped difference of two
//      samples at once, a saturating negate moves "difference > 0" into the sign bits and v_dot2 packs the 8 signs of a
//      lane into a byte (sample order); the byte goes to LDS.  The phase of a lane's ninth sample is its neighbour's first.
//   B  each lane takes one 32-sample word of sign bits plus the two words after it: 18 funnel shifts (the check bits sit
//      two samples apart) shared by both sync words, which are bitwise complements of each other on their first 18 bits,
//      so one AND chain and one OR chain decide both.
// Algorithmic traffic: 2 B per sample, read once (+ 128 B of halo per chunk).
constexpr int kUatScanThreads = 1024, kUatScanWaves = kUatScanThreads / 64, kUatRows = 4;
constexpr int kUatRowSamples  = 64 * 8;
constexpr int kUatWaveSamples = kUatRows * kUatRowSamples;     // 2048
constexpr int kUatChunk       = kUatScanWaves * kUatWaveSamples; // 32768 samples = 1024 sign words
constexpr int kUatChunkWords  = kUatChunk / 32;
constexpr uint32_t kUatParkCap = 1024;
static_assert((0xEACDDA4E2ull >> 18) == (~(0x153225B1Dull >> 18) & 0x3FFFFull), "the two check words are complements");

__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b)
{
    typedef short v2s __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (v2s)(__builtin_bit_cast(v2s, a) - __builtin_bit_cast(v2s, b)));
}
__device__ __forceinline__ uint32_t pk_neg_sat_i16(uint32_t a)
{
    uint32_t r;
    asm("v_pk_sub_i16 %0, 0, %1 clamp" : "=v"(r) : "v"(a));
    return r;
}

// sign byte of 8 consecutive samples whose 9 phases are known: bit k = (phi[k + 1] - phi[k] wrapped to int16) > 0
__device__ __forceinline__ uint32_t sign_byte(uint32_t p01, uint32_t p23, uint32_t p45, uint32_t p67, uint32_t p8)
{
    const uint32_t p12 = __builtin_amdgcn_alignbit(p23, p01, 16), p34 = __builtin_amdgcn_alignbit(p45, p23, 16);
    const uint32_t p56 = __builtin_amdgcn_alignbit(p67, p45, 16), p78 = __builtin_amdgcn_alignbit(p8, p67, 16);
    typedef unsigned short v2u __attribute__((ext_vector_type(2)));
    uint32_t acc = 0;
    // -d saturated: negative exactly when d > 0 (d = -32768 becomes +32767)
    acc = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u, pk_neg_sat_i16(pk_sub_i16(p12, p01)) & 0x80008000u), (v2u){1, 2}, acc, false);
    acc = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u, pk_neg_sat_i16(pk_sub_i16(p34, p23)) & 0x80008000u), (v2u){4, 8}, acc, false);
    acc = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u, pk_neg_sat_i16(pk_sub_i16(p56, p45)) & 0x80008000u), (v2u){16, 32}, acc, false);
    acc = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u, pk_neg_sat_i16(pk_sub_i16(p78, p67)) & 0x80008000u), (v2u){64, 128}, acc, false);
    return acc >> 15;
}

__device__ __forceinline__ uint32_t lut2(const uint16_t* __restrict__ lut_s, uint32_t iq2)
{ // phases of the two samples in one dword, packed the same way
#if defined(UAT_EXP_NO_GATHER)
    (void)lut_s;
    return iq2 * 0x9E3779B1u;
#endif
    return (uint32_t)lut_s[iq2 & 0xFFFFu] | ((uint32_t)lut_s[iq2 >> 16] << 16);
}

// the same byte for 8 samples from s0 on, anywhere relative to the end of the stream.  Branch-free on purpose: the nine
// loads go to clamped addresses so that they are all in flight together, the guards only mask results.  n >= 1.
__device__ __forceinline__ uint8_t sign_byte_guarded(const uint16_t* __restrict__ iq, const uint16_t* __restrict__ lut_s, uint64_t n, uint64_t s0)
{
    uint32_t raw[9], ph[9], byte = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) raw[k] = iq[(s0 + (uint64_t)k < n) ? s0 + (uint64_t)k : n - 1];
#pragma unroll
    for (int k = 0; k < 9; k++) ph[k] = lut_s[raw[k]];
#pragma unroll
    for (int k = 0; k < 8; k++) byte |= (s0 + (uint64_t)k + 1 < n && phi_difference(ph[k], ph[k + 1]) > 0) ? (1u << k) : 0u;
    return (uint8_t)byte;
}

__global__ __launch_bounds__(kUatScanThreads) void uat_scan_iq_kernel(const uint16_t* __restrict__ iq, const uint16_t* __restrict__ lut,
                                                                      uint64_t n, uint32_t* __restrict__ cand, uint32_t cap,
                                                                      uint32_t* __restrict__ count)
{
    __shared__ uint16_t lut_s[65536];
    __shared__ uint32_t sign_words[kUatChunkWords + 4]; // + 64 samples of halo (+ slack)
    // matches are parked here and flushed with ONE global atomic per ~512 of them: appending each match with its own
    // atomicAdd on the shared counter serialises in L2 (measured: 131 k matches per GiB cost 1.4 ms, the scan itself 0.25 ms)
    __shared__ uint32_t parked[kUatParkCap];
    __shared__ uint32_t parked_count, flush_base;
    uint8_t* const      sign_bytes = reinterpret_cast<uint8_t*>(sign_words);
    const int           tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < 65536 / 8; i += kUatScanThreads) reinterpret_cast<uint4*>(lut_s)[i] = reinterpret_cast<const uint4*>(lut)[i];
    if (tid < 4) sign_words[kUatChunkWords + tid] = 0;
    if (tid == 0) parked_count = 0;
    __syncthreads();

    const uint64_t nchunks = (n + kUatChunk - 1) / kUatChunk;
    for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x)
    {
        const uint64_t base = chunk * kUatChunk;
        // ---- A: signs.  Wave w owns samples [base + 2048 w, + 2048); wave 15 also does the halo, the 64 samples after the chunk.
        const uint64_t wave_start = base + (uint64_t)wave * kUatWaveSamples;
        if (wave_start + kUatWaveSamples + 1 <= n)
        { // every difference has both samples: all four 16-byte loads in flight, then 32 gathers
            uint4 v[kUatRows];
#pragma unroll
            for (int r = 0; r < kUatRows; r++) v[r] = *reinterpret_cast<const uint4*>(iq + wave_start + (uint64_t)r * kUatRowSamples + (uint64_t)lane * 8);
            const uint32_t after_wave = iq[wave_start + kUatWaveSamples]; // uniform address
            uint32_t       p01[kUatRows], p23[kUatRows], p45[kUatRows], p67[kUatRows];
#pragma unroll
            for (int r = 0; r < kUatRows; r++)
            {
                p01[r] = lut2(lut_s, v[r].x), p23[r] = lut2(lut_s, v[r].y);
                p45[r] = lut2(lut_s, v[r].z), p67[r] = lut2(lut_s, v[r].w);
            }
#pragma unroll
            for (int r = 0; r < kUatRows; r++)
            { // a lane's ninth phase is the next lane's first (wave_shl:1); lane 63's is the next row's first
                const uint32_t after = (r + 1 < kUatRows) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)p01[(r + 1) % kUatRows]) : (uint32_t)lut_s[after_wave];
                const uint32_t p8    = (uint32_t)__builtin_amdgcn_update_dpp((int)after, (int)p01[r], 0x130, 0xF, 0xF, false);
                sign_bytes[wave * (kUatWaveSamples / 8) + r * 64 + lane] = (uint8_t)sign_byte(p01[r], p23[r], p45[r], p67[r], p8 & 0xFFFFu);
            }
        }
        else
        { // the stream ends in (or before) this wave's span
            for (int r = 0; r < kUatRows; r++)
                sign_bytes[wave * (kUatWaveSamples / 8) + r * 64 + lane] = sign_byte_guarded(iq, lut_s, n, wave_start + (uint64_t)r * kUatRowSamples + (uint64_t)lane * 8);
        }
        if (wave == kUatScanWaves - 1)
        { // halo: the first 64 samples of the next chunk, 8 lanes
            const uint64_t h0 = base + kUatChunk;
            uint32_t       byte;
            if (h0 + 64 + 1 <= n)
            {
                uint4 v = {0, 0, 0, 0};
                if (lane < 8) v = *reinterpret_cast<const uint4*>(iq + h0 + (uint64_t)lane * 8);
                const uint32_t p01 = lut2(lut_s, v.x), p23 = lut2(lut_s, v.y), p45 = lut2(lut_s, v.z), p67 = lut2(lut_s, v.w);
                const uint32_t after = lut_s[iq[h0 + 64]];
                uint32_t       p8    = (uint32_t)__builtin_amdgcn_update_dpp((int)after, (int)p01, 0x130, 0xF, 0xF, false);
                if (lane == 7) p8 = after;
                byte = sign_byte(p01, p23, p45, p67, p8 & 0xFFFFu);
            }
            else byte = sign_byte_guarded(iq, lut_s, n, h0 + (uint64_t)(lane & 7) * 8);
            if (lane < 8) sign_bytes[kUatChunk / 8 + lane] = (uint8_t)byte;
        }
        __syncthreads();
        // ---- B: one word of 32 start positions per lane
#if !defined(UAT_EXP_NO_MATCH)
        {
            const uint32_t w0 = sign_words[tid], w1 = sign_words[tid + 1], w2 = sign_words[tid + 2];
            uint32_t       all = 0xFFFFFFFFu, any = 0u; // over k of "bit k agrees with the ADS-B check word"
#pragma unroll
            for (int k = 0; k < 18; k++)
            {
                const int sh = 2 * k;
                uint32_t  v  = sh == 0 ? w0 : sh < 32 ? __builtin_amdgcn_alignbit(w1, w0, sh) : sh == 32 ? w1 : __builtin_amdgcn_alignbit(w2, w1, sh - 32);
                if (!((kAdsbSync >> (35 - k)) & 1ull)) v = ~v;
                all &= v;
                any |= v;
            }
            uint32_t hits = all | ~any; // ADS-B word: every bit agrees; uplink word: none does
            if (hits)
            {
                const uint64_t word_start = base + (uint64_t)tid * 32;
                while (hits)
                {
                    const int j = __builtin_ctz(hits);
                    hits &= hits - 1;
                    const uint64_t i = word_start + (uint64_t)j;
                    if (i + 36 > n) continue;
                    const uint32_t kind  = ((all >> j) & 1u) ? 0u : 1u;
                    const uint32_t value = ((uint32_t)i & 0x7FFFFFFFu) | (kind << 31);
                    const uint32_t at    = atomicAdd(&parked_count, 1u);
                    if (at < kUatParkCap) parked[at] = value;
                    else
                    { // more matches in flight than the parking area holds: straight to the global list
                        const uint32_t slot = atomicAdd(count, 1u);
                        if (slot < cap) cand[slot] = value;
                    }
                }
            }
        }
#endif
        __syncthreads();
        const uint32_t pending = parked_count < kUatParkCap ? parked_count : kUatParkCap; // same for every lane
        if (pending >= kUatParkCap / 2 || (pending && chunk + gridDim.x >= nchunks))
        {
            if (tid == 0) flush_base = atomicAdd(count, pending);
            __syncthreads();
            for (uint32_t k = tid; k < pending; k += kUatScanThreads)
                if (flush_base + k < cap) cand[flush_base + k] = parked[k];
            __syncthreads();
            if (tid == 0) parked_count = 0;
        }
    }
}

// ---- K3: one wave per candidate
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

// sign bits of the 64 samples from p on (bit 0 = sample p; 0 where the difference needs a sample beyond the stream)
template <bool PHASES_GIVEN>
__device__ __forceinline__ uint64_t sign_window(const uint16_t* __restrict__ in, const uint16_t* __restrict__ lut, uint64_t n, uint64_t p, int lane)
{
    return __ballot(dphi_at<PHASES_GIVEN>(in, lut, n, p + (uint64_t)lane) > 0);
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
                                                       const uint32_t* __restrict__ cand, uint32_t ncand, uat_adsb_rec_t* __restrict__ adsb,
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
            const uint64_t sb = idx >> 1;
            const uint64_t w0 = sign_window<PHASES_GIVEN>(in, lut, n, 2 * sb, lane);
            const uint64_t w1 = sign_window<PHASES_GIVEN>(in, lut, n, 2 * (sb + 36 + 240 + 1), lane);
            const uint64_t w2 = sign_window<PHASES_GIVEN>(in, lut, n, 2 * (sb + 36 + 384 + 1), lane);
            if (lane == 0)
            {
                r->index    = (uint32_t)idx;
                r->kind     = 0;
                r->window   = w0;
                r->after[0] = w1;
                r->after[1] = w2;
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
            const uint64_t sb = idx >> 1;
            const uint64_t w0 = sign_window<PHASES_GIVEN>(in, lut, n, 2 * sb, lane);
            const uint64_t w1 = sign_window<PHASES_GIVEN>(in, lut, n, 2 * (sb + 36 + 4416 + 1), lane);
            if (lane == 0)
            {
                adsb[c].index    = (uint32_t)idx;
                adsb[c].kind     = 1;
                adsb[c].window   = w0;
                adsb[c].after[0] = w1;
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
    if (!a.phases_given)
    {
        const uint64_t nchunks = (a.nsamples + kUatChunk - 1) / kUatChunk;
#if !defined(UAT_EXP_GRID)
#define UAT_EXP_GRID 256
#endif
        const uint32_t grid    = (uint32_t)(nchunks < UAT_EXP_GRID ? nchunks : UAT_EXP_GRID);
        hipLaunchKernelGGL(uat_scan_iq_kernel, dim3(grid), dim3(kUatScanThreads), 0, stream, a.in, a.lut, a.nsamples, a.cand, a.cand_cap, a.counts);
        return hipGetLastError();
    }
    const uint64_t nwords = (a.nsamples + 63) / 64;
    e = hipMemsetAsync(a.signs + nwords, 0, 2 * sizeof(uint64_t), stream);
    if (e != hipSuccess) return e;
    uint32_t       g1     = (uint32_t)((nwords + 3) / 4 > 8192 ? 8192 : (nwords + 3) / 4);
    hipLaunchKernelGGL(uat_sign_kernel, dim3(g1), dim3(256), 0, stream, a.in, a.nsamples, a.signs);
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
        hipLaunchKernelGGL(uat_demod_kernel<true>, dim3(g), dim3(64), 0, stream, a.in, a.lut, a.nsamples, a.cand, ncand, a.adsb, a.uplink, a.uplink_cap,
                           a.counts + 1);
    else
        hipLaunchKernelGGL(uat_demod_kernel<false>, dim3(g), dim3(64), 0, stream, a.in, a.lut, a.nsamples, a.cand, ncand, a.adsb, a.uplink, a.uplink_cap,
                           a.counts + 1);
    return hipGetLastError();
}

} // namespace adsb_amd
