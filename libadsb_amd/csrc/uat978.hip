// uat978.hip -- gfx950 kernels for the UAT 978 path: u8 IQ -> phase (reference LUT) -> sign of the phase difference ->
// 18-bit sync search on both sample alignments -> per-candidate sync re-check and frame slicing.
//
// What comes from the reference tree: the phase LUT (UAT978.cpp:76-100) and the map phi = lut[I | Q << 8] (:52).
// Everything after that restates the published dump978 legacy demodulator (un-vendored in the reference):
// parity unpinned, GPU == oracle/oracle978.c is what the tests check.
//
// Kernels over a device-resident stream:
//   uat_scan_iq_kernel     batch path (u8 IQ in HBM): persistent 1024-lane workgroups, the whole 128 KiB phase LUT in LDS
//                          (bank-swizzled), wave-owned 2048-sample spans with a register prefetch of the next span: phases,
//                          wrapped 16-bit differences, sign bits, both 18-bit check words -- bound by the HBM read (2 B/sample)
//   uat_sign_kernel +      the same two steps for a buffer of *phases* (the process_buffer seam and the 65 536-entry staging
//   uat_match_kernel       rounds of HandleData; small inputs)
//   uat_demod_kernel       one wave per match (the host may also ask for further sample indices, see uat978_host.cpp), twelve waves to a
//                          workgroup that shares one quadrant of the phase LUT in LDS (the table's symmetries give the rest exactly):
//                          one coalesced burst stages the phase differences of the match's samples in LDS (ten tiles for an uplink frame), then
//                          the 36-bit sync re-check against the data-derived centre for the match and, when that one needed
//                          corrections, for the next sample (the reference tries both and keeps the better), the frame sliced a
//                          byte per lane at that centre, its syndromes with the wave across the symbols and its Reed-Solomon
//                          decode with the whole wave on one code word (rs978.h: the locator by elimination for the two ADS-B codes,
//                          Berlekamp-Massey otherwise, Chien, Forney with libfec's conventions); the choice between the two alignments is made here; and the frames the scan loop would take
//                          behind this one through stale register bits (StaleWindow), by the same wave.  A record is 32 bytes, the
//                          corrected ADS-B frame bytes go to a parallel array
//   uat_order_*            counting sort of the matches by stream position, on the device; also lists the uplink matches, which the
//                          demodulation takes first
//   uat_succ_kernel +      which frames the dump978 scan loop takes: a successor function over the ordered matches, the path from the
//   uat_mark_kernel        first one marked by pointer jumping in blocks of 4096 matches
// The host walks the frames taken (uat978_host.cpp).
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "diag.hip.h"
#include "rs978.h"
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

__device__ __forceinline__ void wave_lds_fence()
{ // orders this wave's LDS traffic; no other wave touches the same words
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
        accA &= ~((accA << 1) & 0xAAAAAAAAu), accU &= ~((accU << 1) & 0xAAAAAAAAu); // odd twin of an even match: never the loop's choice (see uat_scan_iq_kernel)
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
// the index, UAT978.cpp:52), shared read-only; everything else is private to a wave, so there is no barrier after the
// table load.  A wave owns whole 2 048-sample spans (span s -> wave s mod nwaves):
//   A  4 rows of 512 samples, 16 B (8 samples) per lane per row, plus 64 samples of halo on 8 lanes; the loads of the next
//      span are issued before this one is worked on.  8 LDS gathers per lane per row give the phases as packed u16 pairs,
//      the pair shifted by one sample comes from v_alignbit, v_pk_sub_i16 is the wrapped difference of two samples at
//      once, a saturating negate moves "difference > 0" into the sign bits and v_dot2 packs the 8 signs of a lane into a
//      byte (sample order); the byte goes to the wave's corner of LDS.  A lane's ninth phase is its neighbour's first.
//   B  each lane takes one 32-sample word of sign bits plus the two words after it: 18 funnel shifts (the check bits sit
//      two samples apart) shared by both sync words, which are bitwise complements of each other on their first 18 bits,
//      so one AND chain and one OR chain decide both.
// Algorithmic traffic: 2 B per sample, read once (+ 128 B of halo per 4 KiB span, which the neighbouring wave reads anyway).
constexpr int kUatScanThreads = 1024, kUatScanWaves = kUatScanThreads / 64, kUatRows = 4;
constexpr int kUatRowSamples  = 64 * 8;
constexpr int kUatWaveSamples = kUatRows * kUatRowSamples;     // 2048
constexpr int kUatSpanWords   = kUatWaveSamples / 32; // 64: one per lane
constexpr uint32_t kUatParkCap = 64;  // per wave
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

// LDS bank of a LUT entry = bits 1..6 of its index = bits 1..6 of I alone (Q only moves the address by multiples of 512 B),
// so samples that differ in Q but not in I pile onto one bank; receiver noise (I, Q within a few LSB of 127) uses ~4 of the
// 64 banks.  The table is therefore stored with Q's low six bits XORed into those index bits (a bijection), and every
// look-up applies the same XOR: one extra shift-and-mask and one XOR per two samples.
__device__ __forceinline__ uint32_t swz2(uint32_t iq2) { return iq2 ^ ((iq2 >> 7) & 0x007E007Eu); } // both u16 halves at once
__device__ __forceinline__ uint32_t swz1(uint32_t iq) { return swz2(iq) & 0xFFFFu; }

// byte offset into the u16 table of the sample in the low / high half of `y`: one SDWA shift each (the compiler would
// otherwise mask and shift separately, and fold the swizzle XOR into both halves' address computations)
__device__ __forceinline__ uint32_t table_offset_lo(uint32_t y)
{
    uint32_t r;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(1u), "v"(y));
    return r;
}
__device__ __forceinline__ uint32_t table_offset_hi(uint32_t y)
{
    uint32_t r;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(1u), "v"(y));
    return r;
}

// a match word into the bin of its stretch of the stream (see uat_scan_iq_kernel); counts = the call's UatCount words
__device__ __forceinline__ void place_in_bin(uint32_t* __restrict__ bin_fill, uint32_t* __restrict__ bin_slots, uint32_t* __restrict__ counts, uint32_t value)
{
    const uint32_t bin = (value & 0x7FFFFFFFu) >> kUatBinShift;
    const uint32_t at  = atomicAdd(&bin_fill[bin], 1u);
    if (at < kUatBinCap) bin_slots[(size_t)bin * kUatBinCap + at] = value;
    else atomicOr(&counts[kUatCountBinOverflow], 1u);
}

__device__ __forceinline__ uint32_t lut2(const uint16_t* __restrict__ lut_s, uint32_t iq2)
{ // phases of the two samples in one dword, packed the same way
    const uint32_t y  = swz2(iq2);
    const char*    b  = reinterpret_cast<const char*>(lut_s);
    const uint32_t lo = *reinterpret_cast<const uint16_t*>(b + table_offset_lo(y));
    const uint32_t hi = *reinterpret_cast<const uint16_t*>(b + table_offset_hi(y));
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u); // hi.word0 : lo.word0
}

// the same byte for 8 samples from s0 on, anywhere relative to the end of the stream.  Branch-free on purpose: the nine
// loads go to clamped addresses so that they are all in flight together, the guards only mask results.  n >= 1.
__device__ __forceinline__ uint8_t sign_byte_guarded(const uint16_t* __restrict__ iq, const uint16_t* __restrict__ lut_s, uint64_t n, uint64_t s0)
{
    uint32_t raw[9], ph[9], byte = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) raw[k] = iq[(s0 + (uint64_t)k < n) ? s0 + (uint64_t)k : n - 1];
#pragma unroll
    for (int k = 0; k < 9; k++) ph[k] = lut_s[swz1(raw[k])];
#pragma unroll
    for (int k = 0; k < 8; k++) byte |= (s0 + (uint64_t)k + 1 < n && phi_difference(ph[k], ph[k + 1]) > 0) ? (1u << k) : 0u;
    return (uint8_t)byte;
}

// bin_fill / bin_slots (round 6; null: not kept): besides the list, every match is placed in the bin of its 32 768-sample stretch of the stream --
// kUatBinCap slots per bin, handed out by an atomic on the bin's fill count -- so that the ordering is ONE launch afterwards (uat_order_bins_kernel:
// a prefix over the fill counts and a sorting network per bin) instead of the four of the counting sort over the list.  A bin that runs over
// (dense noise that looks like check words, a constructed input) raises count[kUatCountBinOverflow]; the host then orders the list the old way.
__global__ __launch_bounds__(kUatScanThreads) void uat_scan_iq_kernel(const uint16_t* __restrict__ iq, const uint16_t* __restrict__ lut,
                                                                      uint64_t n, uint32_t* __restrict__ cand, uint32_t cap,
                                                                      uint32_t* __restrict__ count, uint32_t* __restrict__ bin_fill,
                                                                      uint32_t* __restrict__ bin_slots)
{
    // The workgroup shares only the (read-only) table.  Each wave owns whole 2 048-sample spans: it produces the span's sign
    // bits plus 64 samples of halo into its own corner of LDS and searches them itself, so after the table is loaded no
    // barrier is needed and the sixteen waves of a CU drift apart freely (loads of one under gathers and ALU of the others).
    __shared__ uint16_t lut_s[65536];
    __shared__ uint32_t sign_words[kUatScanWaves][kUatSpanWords + 4]; // + 2 halo words (+ slack)
    // matches are parked per wave and flushed with ONE global atomic per ~32 of them: appending each match with its own
    // atomicAdd on the shared counter serialises in L2 (measured: 131 k matches per GiB cost 1.4 ms, the scan itself 0.25 ms)
    __shared__ uint32_t parked[kUatScanWaves][kUatParkCap];
    __shared__ uint32_t parked_count[kUatScanWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < 65536 / 2; i += kUatScanThreads)
    { // two entries per lane per trip; they land swizzled
        const uint32_t two = reinterpret_cast<const uint32_t*>(lut)[i];
        lut_s[swz1(2u * (uint32_t)i)]      = (uint16_t)two;
        lut_s[swz1(2u * (uint32_t)i + 1u)] = (uint16_t)(two >> 16);
    }
    if (lane < 4) sign_words[wave][kUatSpanWords + lane] = 0;
    if (lane == 0) parked_count[wave] = 0;
    __syncthreads();

    uint32_t* const my_words  = sign_words[wave];
    uint8_t* const  my_bytes  = reinterpret_cast<uint8_t*>(my_words);
    uint32_t* const my_parked = parked[wave];
    const uint64_t  nspans    = (n + kUatWaveSamples - 1) / kUatWaveSamples;
    const uint64_t  nwaves    = (uint64_t)gridDim.x * kUatScanWaves;

    auto flush = [&](uint32_t pending)
    { // wave-wide; `pending` <= kUatParkCap entries of my_parked go to the global list
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(count, pending);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        for (uint32_t k = lane; k < pending; k += 64)
        {
            const uint32_t value = my_parked[k];
            if (base + k < cap) cand[base + k] = value;
            if (bin_fill) place_in_bin(bin_fill, bin_slots, count, value);
        }
        wave_lds_fence();
        if (lane == 0) parked_count[wave] = 0;
        wave_lds_fence();
    };

    // span s of the stream belongs to wave (s mod nwaves): neighbouring waves read neighbouring 4 KiB
    uint64_t span = (uint64_t)blockIdx.x * kUatScanWaves + (uint64_t)wave;
    // register prefetch: the loads of the next span are issued before this one is worked on
    uint4    v[kUatRows], vh = {0, 0, 0, 0};
    uint32_t after_halo = 0;
    auto     fast       = [&](uint64_t sp) { return sp < nspans && (sp + 1) * kUatWaveSamples + 64 + 1 <= n; };
    auto     load       = [&](uint64_t sp)
    {
        const uint64_t s0 = sp * kUatWaveSamples;
#pragma unroll
        for (int r = 0; r < kUatRows; r++) v[r] = *reinterpret_cast<const uint4*>(iq + s0 + (uint64_t)r * kUatRowSamples + (uint64_t)lane * 8);
        vh         = *reinterpret_cast<const uint4*>(iq + s0 + kUatWaveSamples + (uint64_t)(lane & 7) * 8); // halo: lanes 0..7 matter
        after_halo = iq[s0 + kUatWaveSamples + 64];
    };
    if (fast(span)) load(span);
    for (; span < nspans; span += nwaves)
    {
        const uint64_t s0 = span * kUatWaveSamples;
        // ---- A: sign bits of [s0, s0 + 2048 + 64)
        __builtin_amdgcn_s_setprio(0);
        if (fast(span))
        {
            uint32_t p01[kUatRows + 1], p23[kUatRows + 1], p45[kUatRows + 1], p67[kUatRows + 1];
#pragma unroll
            for (int r = 0; r < kUatRows; r++)
            {
                p01[r] = lut2(lut_s, v[r].x), p23[r] = lut2(lut_s, v[r].y);
                p45[r] = lut2(lut_s, v[r].z), p67[r] = lut2(lut_s, v[r].w);
            }
            p01[kUatRows] = lut2(lut_s, vh.x), p23[kUatRows] = lut2(lut_s, vh.y);
            p45[kUatRows] = lut2(lut_s, vh.z), p67[kUatRows] = lut2(lut_s, vh.w);
            const uint32_t after_all = lut_s[swz1(after_halo)];
            if (fast(span + nwaves)) load(span + nwaves); // everything of this span is in registers as phases now
            // From here to the end of the span the wave has everything in registers or in its own corner of LDS; with raised priority it
            // gets through that part ahead of the waves that are still issuing table gathers, and is back at issuing its own sooner:
            // 0.2068 -> 0.2007 ms per GiB (in-process A/B; raising it for the gathers instead gives 0.2040, for the search alone 0.2055).
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int r = 0; r <= kUatRows; r++)
            { // a lane's ninth phase is the next lane's first (wave_shl:1); lane 63's is the next row's first (row 4 = halo, 8 lanes)
                const uint32_t after = (r < kUatRows) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)p01[(r + 1) % (kUatRows + 1)]) : after_all;
                uint32_t       p8    = (uint32_t)__builtin_amdgcn_update_dpp((int)after, (int)p01[r], 0x130, 0xF, 0xF, false);
                if (r == kUatRows && lane == 7) p8 = after_all;
                const uint32_t byte = sign_byte(p01[r], p23[r], p45[r], p67[r], p8 & 0xFFFFu);
                if (r < kUatRows || lane < 8) my_bytes[r * 64 + lane] = (uint8_t)byte;
            }
        }
        else
        { // the stream ends in (or right after) this span
            for (int r = 0; r < kUatRows; r++) my_bytes[r * 64 + lane] = sign_byte_guarded(iq, lut_s, n, s0 + (uint64_t)r * kUatRowSamples + (uint64_t)lane * 8);
            if (lane < 8) my_bytes[kUatRows * 64 + lane] = sign_byte_guarded(iq, lut_s, n, s0 + kUatWaveSamples + (uint64_t)lane * 8);
            if (fast(span + nwaves)) load(span + nwaves);
        }
        wave_lds_fence();
        // ---- B: one word of 32 start positions per lane
        {
            const uint32_t w0 = my_words[lane], w1 = my_words[lane + 1], w2 = my_words[lane + 2];
            uint32_t       all = 0xFFFFFFFFu, any = 0u; // over k of "bit k agrees with the ADS-B check word"
#pragma unroll
            for (int k = 0; k < 18; k++)
            {
                const int sh = 2 * k;
                uint32_t  x  = sh == 0 ? w0 : sh < 32 ? __builtin_amdgcn_alignbit(w1, w0, sh) : sh == 32 ? w1 : __builtin_amdgcn_alignbit(w2, w1, sh - 32);
                if (!((kAdsbSync >> (35 - k)) & 1ull)) x = ~x;
                all &= x;
                any |= x;
            }
            // A check word that matches on the even sample of a bit time and again on the odd one: the scan loop looks at register 0
            // first and never takes the second match, so it is dropped here (a word starts on an even sample: the twin is the next
            // bit of the same mask).  Only a look-up after a jump can still ask for it; the host has it demodulated on demand then.
            uint32_t none = ~any;
            all &= ~((all << 1) & 0xAAAAAAAAu), none &= ~((none << 1) & 0xAAAAAAAAu);
            uint32_t hits = all | none; // ADS-B word: every bit agrees; uplink word: none does
            if (hits)
            {
                const uint64_t word_start = s0 + (uint64_t)lane * 32;
                while (hits)
                {
                    const int j = __builtin_ctz(hits);
                    hits &= hits - 1;
                    const uint64_t i = word_start + (uint64_t)j;
                    if (i + 36 > n) continue;
                    const uint32_t kind  = ((all >> j) & 1u) ? 0u : 1u;
                    const uint32_t value = ((uint32_t)i & 0x7FFFFFFFu) | (kind << 31);
                    const uint32_t at    = atomicAdd(&parked_count[wave], 1u);
                    if (at < kUatParkCap) my_parked[at] = value;
                    else
                    { // more matches in one span than the parking area holds: straight to the global list
                        const uint32_t slot = atomicAdd(count, 1u);
                        if (slot < cap) cand[slot] = value;
                        if (bin_fill) place_in_bin(bin_fill, bin_slots, count, value);
                    }
                }
            }
        }
        wave_lds_fence();
        const uint32_t pending = parked_count[wave] < kUatParkCap ? parked_count[wave] : kUatParkCap; // same for every lane
        if (pending >= kUatParkCap / 2) flush(pending);
    }
    wave_lds_fence();
    const uint32_t pending = parked_count[wave] < kUatParkCap ? parked_count[wave] : kUatParkCap;
    if (pending) flush(pending);
}

// ---- K3: one wave per candidate: sync re-check, slicing, Reed-Solomon, for the candidate sample and the next one
struct SyncCheck
{
    bool ok;
    int  center;
};

template <bool PHASES_GIVEN>
__device__ __forceinline__ int dphi_at(const uint16_t* __restrict__ in, const uint16_t* __restrict__ lut, uint64_t n, uint64_t s)
{
    if (s + 1 >= n) return 0;
    const uint32_t a = PHASES_GIVEN ? in[s] : lut[in[s]];
    const uint32_t b = PHASES_GIVEN ? in[s + 1] : lut[in[s + 1]];
    return phi_difference(a, b);
}

// sign bits of the 64 samples from p on, split by sample alignment the way the two shift registers see them: bit k of the
// low word = sample p + 2k (register 0 when p is even), bit k of the high word = sample p + 1 + 2k (register 1); 0 where
// the difference needs a sample beyond the stream
template <bool PHASES_GIVEN>
__device__ __forceinline__ uint64_t sign_window(const uint16_t* __restrict__ in, const uint16_t* __restrict__ lut, uint64_t n, uint64_t p, int lane)
{
    const uint64_t s = p + 2ull * (uint64_t)(lane & 31) + (uint64_t)(lane >> 5);
    return __ballot(dphi_at<PHASES_GIVEN>(in, lut, n, s) > 0);
}

// check_sync_word: centre = mean of the per-class means of dphi over the 36 sync bits (C integer division), then at most
// four bits on the wrong side of it
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

// slice `nbits` frame bits starting at sample `start` (first bit after the sync word), MSB-first bytes into out[] (LDS).
// Eight groups of 64 bits per trip so that their (dependent: sample, then LUT) loads are in flight together.
template <bool PHASES_GIVEN>
__device__ __forceinline__ void slice_frame(const uint16_t* __restrict__ in, const uint16_t* __restrict__ lut, uint64_t n, uint64_t start,
                                            int center, int nbits, uint8_t* out, int lane)
{
    constexpr int kGroups = 8;
    for (int base = 0; base < nbits; base += 64 * kGroups)
    {
        uint32_t a[kGroups], b[kGroups];
#pragma unroll
        for (int u = 0; u < kGroups; u++)
        {
            const int      bit = base + u * 64 + lane;
            const uint64_t s   = start + 2ull * (uint64_t)bit;
            const bool     in_range = bit < nbits && s + 1 < n;
            a[u] = in[in_range ? s : 0], b[u] = in[in_range ? s + 1 : 0];
        }
        if (!PHASES_GIVEN)
        {
#pragma unroll
            for (int u = 0; u < kGroups; u++) a[u] = lut[a[u]], b[u] = lut[b[u]];
        }
#pragma unroll
        for (int u = 0; u < kGroups; u++)
        {
            const int      gbase = base + u * 64, bit = gbase + lane;
            const uint64_t s     = start + 2ull * (uint64_t)bit;
            const int      d     = (bit < nbits && s + 1 < n) ? phi_difference(a[u], b[u]) : 0;
            const uint64_t bits  = __ballot(bit < nbits && d > center); // bit `lane` = frame bit gbase + lane
            if (lane < 8 && gbase + 8 * lane < nbits)
            { // byte k of this group = frame bits gbase + 8k .. 8k + 7, first bit = MSB
                const uint32_t byte      = (uint32_t)(bits >> (8 * lane)) & 0xFFu;
                out[(gbase >> 3) + lane] = (uint8_t)(__builtin_bitreverse32(byte) >> 24);
            }
        }
    }
}

// ---- the demodulating wave works from a tile of phase differences in LDS: entry k = dphi(base + k) for k < kUatTileValid, zero
// where the difference needs a sample beyond the stream.  One coalesced burst of 16-byte loads (two per lane) and, for IQ
// input, sixteen independent LUT gathers per lane fill it; everything after that -- sign windows, both sync re-checks, both
// slicings -- reads LDS instead of chasing sample -> LUT -> sample + 1 -> LUT through global memory per step.
constexpr int kUatTile      = 1024; // samples staged (128 chunks of 8)
constexpr int kUatTileValid = 1023; // the last entry would need the first phase of the next tile
constexpr int kUatTileStride = 896; // uplink frames: tile t starts 896 t samples after the first (7 groups of 64 bits)

// Out of line (two call sites), so the address spaces of its operands are spelled out: as generic pointers the stream and the table were
// read by flat loads, which count against the LDS counter as well and are waited for with everything else.
//   counter  null, or the work counter to draw the wave's next ticket from.  The returning atomic is issued BEHIND the tile's table
//            look-ups and waited for together with them, so that its round trip to the memory side (device scope: past the L2) is not
//            a wait of its own at the head of every match.  Returns the ticket (uniform), 0xFFFFFFFF without a counter.
typedef __attribute__((address_space(1))) const uint16_t* g_cu16;
typedef __attribute__((address_space(1))) uint32_t*       g_u32;
typedef __attribute__((address_space(3))) int16_t*        lds_i16;

// ---- The phase table in LDS, folded (round 5).  The demodulating waves used to gather their tiles' phases from the 128 KB table in
// global memory: sixteen 64-lane gathers per tile, every lane on a cache line of its own wherever a frame is on the air (its samples go round
// the IQ plane), each costing the CU's address unit ~64 cycles -- 0.08 of this kernel's 0.21 ms for the ADS-B matches' first tiles alone
// (profiles/r05_uat978_demod_parts.txt: staging with the look-ups 0.124 ms, without 0.042) and as much again for the uplink frames' ten tiles.
// The table is atan2 about (127.5, 127.5), so it has the plane's symmetries EXACTLY (checked entry by entry when a handle is made,
// uat978_host.cpp, and in tests/test_uat978.py): with I' = 255 - I, Q' = 255 - Q
//     lut(I', Q) = 32768 - lut(I, Q),    lut(I, Q') = -lut(I, Q)      (mod 65536)
// and the quadrant I, Q >= 128 -- 128 x 128 entries, 32 KB -- is enough: a workgroup of kUatDemodWaves waves shares one copy in LDS.
constexpr int kUatFoldedEntries = 128 * 128;
typedef __attribute__((address_space(3))) const uint16_t* lds_cu16;

// phases of the two samples in `w` (packed the same way) from the folded table
__device__ __forceinline__ uint32_t lut2_folded(lds_cu16 folded, uint32_t w)
{
    typedef short          v2s __attribute__((ext_vector_type(2)));
    typedef unsigned short v2u __attribute__((ext_vector_type(2)));
    const uint32_t sgn = w & 0x80808080u;                 // per byte: the coordinate is >= 128
    const uint32_t m   = w ^ (0x7F7F7F7Fu + (sgn >> 7));  // b >= 128: b - 128, else 127 - b (seven bits per byte)
    // byte offset of entry (mq, mi) = mq << 8 | mi << 1, in both halves at once
    uint32_t       o2  = (m & 0xFF00FF00u) | ((m << 1) & 0x00FF00FFu);
    // The bank of an entry is bits 2..7 of its offset = mi's bits 1..6 alone: the samples of a frame (a circle in the IQ plane: |I - 127.5| near
    // the amplitude for much of the way round) would pile onto a few banks.  Row mq is therefore stored with mq's low six bits XORed into them.
    o2 ^= (o2 >> 6) & 0x00FC00FCu;
    const auto     b   = reinterpret_cast<__attribute__((address_space(3))) const char*>(folded);
    const uint32_t lo  = *reinterpret_cast<lds_cu16>(b + (o2 & 0xFFFFu));
    const uint32_t hi  = *reinterpret_cast<lds_cu16>(b + (o2 >> 16));
    uint32_t       ph  = __builtin_amdgcn_perm(hi, lo, 0x05040100u); // hi.word0 : lo.word0
    const uint32_t isg = sgn << 8;                                   // bit 15 of a half: I >= 128 (bit 15 of sgn: Q >= 128)
    const uint32_t neg = __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2s, sgn ^ isg) >> (v2s){15, 15}); // exactly one coordinate folded: -phase
    ph                 = __builtin_bit_cast(uint32_t, (v2u)(__builtin_bit_cast(v2u, ph ^ neg) - __builtin_bit_cast(v2u, neg)));
    return ph ^ (~isg & 0x80008000u);                                // I folded: + 32768
}

__device__ __forceinline__ uint32_t draw_ticket(g_u32 counter, int lane)
{ // one statement = issue + wait: the compiler does not track the counter of an asm's load, so the result must be there when the asm ends
    uint32_t t = 0;
    if (lane == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(counter), "v"(1u) : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); // (HIP's uint4 is a class: no assignment across address spaces)
typedef __attribute__((address_space(1))) const u32x4* g_cu4;
typedef __attribute__((address_space(3))) u32x4*       lds_u4;

// A tile whose 1024 samples and the sample after them lie inside the stream, as loaded (lane l: samples 8 l .. 8 l + 7 in x0, 8 (l + 64) .. in x1),
// to phase differences in LDS: two phases per register, the pair shifted by one sample from v_alignbit, two wrapped differences per v_pk_sub_i16.
template <bool PHASES_GIVEN>
__device__ __forceinline__ void tile_to_lds(u32x4 x0, u32x4 x1, uint32_t after, lds_cu16 folded, lds_i16 dphi_s, int lane)
{
    const u32x4 x[2] = {x0, x1};
    uint32_t    p[2][4];
#pragma unroll
    for (int r = 0; r < 2; r++)
    {
        const uint32_t w[4] = {x[r].x, x[r].y, x[r].z, x[r].w};
#pragma unroll
        for (int k = 0; k < 4; k++) p[r][k] = PHASES_GIVEN ? w[k] : lut2_folded(folded, w[k]);
    }
    const uint32_t after_ph = PHASES_GIVEN ? after : lut2_folded(folded, after) & 0xFFFFu; // (only lane 63 of round 1 uses it)
#pragma unroll
    for (int r = 0; r < 2; r++)
    {
        const uint32_t wrap = r == 0 ? (uint32_t)__builtin_amdgcn_readlane((int)p[1][0], 0) : after_ph;
        const uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp((int)wrap, (int)p[r][0], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
        const uint32_t p12 = __builtin_amdgcn_alignbit(p[r][1], p[r][0], 16), p34 = __builtin_amdgcn_alignbit(p[r][2], p[r][1], 16);
        const uint32_t p56 = __builtin_amdgcn_alignbit(p[r][3], p[r][2], 16), p78 = __builtin_amdgcn_alignbit(next, p[r][3], 16);
        *reinterpret_cast<lds_u4>(dphi_s + 8 * (lane + 64 * r)) =
            u32x4{pk_sub_i16(p12, p[r][0]), pk_sub_i16(p34, p[r][1]), pk_sub_i16(p56, p[r][2]), pk_sub_i16(p78, p[r][3])};
    }
}
// (out of line for the registers, like stage_dphi: the caller loaded the tile itself, while it was still busy with the tile before)
template <bool PHASES_GIVEN>
__device__ __noinline__ void stage_loaded_tile(u32x4 x0, u32x4 x1, uint32_t after, lds_cu16 folded, lds_i16 dphi_s, int lane)
{
    folded = (lds_cu16)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)folded);
    dphi_s = (lds_i16)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)dphi_s);
    tile_to_lds<PHASES_GIVEN>(x0, x1, after, folded, dphi_s, lane);
}

template <bool PHASES_GIVEN>
__device__ __forceinline__ uint32_t stage_dphi_body(g_cu16 in, g_cu16 lut, lds_cu16 folded, uint64_t n, uint64_t base, lds_i16 dphi_s, int lane, g_u32 counter)
{
    uint32_t   ticket  = 0xFFFFFFFFu;
    // the arguments of a function arrive in vector registers; the stream, the table and the counter are the same for every lane: as scalar
    // bases the sixteen look-ups need a 32-bit offset register each instead of a 64-bit address (what the caller may keep live across
    // the call is what this function leaves untouched)
    auto uniform64 = [](uint64_t v) { return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v); };
    in = (g_cu16)uniform64((uint64_t)in), lut = (g_cu16)uniform64((uint64_t)lut), counter = (g_u32)uniform64((uint64_t)counter);
    base = uniform64(base), n = uniform64(n);
    folded = (lds_cu16)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)folded);
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15u) == 0;
    if (aligned && base + (uint64_t)kUatTile + 1 <= n)
    { // the whole tile and the sample after it lie inside the stream (wave-uniform; every tile but a stream's last few): no guards
        const u32x4    x0 = *reinterpret_cast<g_cu4>(in + base + 8ull * (uint64_t)lane), x1 = *reinterpret_cast<g_cu4>(in + base + 8ull * (uint64_t)(lane + 64));
        const uint32_t after = in[base + (uint64_t)kUatTile];
        if (counter) ticket = draw_ticket(counter, lane);
        tile_to_lds<PHASES_GIVEN>(x0, x1, after, folded, dphi_s, lane);
        return ticket;
    }
    if (counter) ticket = draw_ticket(counter, lane);
    uint32_t ph[2][8];
#pragma unroll
    for (int r = 0; r < 2; r++)
    {
        const uint64_t s = base + 8ull * (uint64_t)(lane + 64 * r);
        uint32_t       raw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) raw[k] = (s + (uint64_t)k < n) ? in[s + (uint64_t)k] : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) ph[r][k] = PHASES_GIVEN ? raw[k] : (uint32_t)lut[raw[k]];
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
    {
        const uint64_t s = base + 8ull * (uint64_t)(lane + 64 * r);
        // the phase after a lane's eighth: the next lane's first; lane 63 of round 0 continues in lane 0 of round 1
        const uint32_t wrap = r == 0 ? (uint32_t)__builtin_amdgcn_readlane((int)ph[1][0], 0) : 0u;
        const uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp((int)wrap, (int)ph[r][0], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
        uint32_t       d[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const uint32_t to = k < 7 ? ph[r][k + 1] : next;
            d[k]              = (s + (uint64_t)k + 1 < n) ? ((to - ph[r][k]) & 0xFFFFu) : 0u;
        }
        *reinterpret_cast<lds_u4>(dphi_s + 8 * (lane + 64 * r)) = u32x4{d[0] | d[1] << 16, d[2] | d[3] << 16, d[4] | d[5] << 16, d[6] | d[7] << 16};
    }
    return ticket;
}

template <bool PHASES_GIVEN>
__device__ __noinline__ uint32_t stage_dphi(g_cu16 in, g_cu16 lut, lds_cu16 folded, uint64_t n, uint64_t base, lds_i16 dphi_s, int lane, g_u32 counter)
{
    return stage_dphi_body<PHASES_GIVEN>(in, lut, folded, n, base, dphi_s, lane, counter);
}

// the sync re-check, the sign windows and the slicing on the staged tile; `off` = the first sample's index inside the tile
__device__ __forceinline__ SyncCheck check_sync_tile(const int16_t* dphi_s, int off, bool uplink, int lane)
{
    static_assert(__builtin_popcountll(kAdsbSync) == 20 && __builtin_popcountll(kUplinkSync) == 16, "bits per class of the sync words");
    const uint64_t pattern = uplink ? kUplinkSync : kAdsbSync;
    const bool     in_sync = lane < 36;
    const int      d       = in_sync ? (int)dphi_s[off + 2 * lane] : 0;
    const bool     one     = in_sync && ((pattern >> ((35 - lane) & 63)) & 1ull);
    const bool     zero    = in_sync && !one;
    const int      one_tot = wave_sum_i(one ? d : 0), zero_tot = wave_sum_i(zero ? d : 0);
    // C integer division by the class sizes (20 ones / 16 zeros in the ADS-B word, the reverse in the uplink word): constants, so
    // no division sequence is emitted
    const int one_mean = uplink ? one_tot / 16 : one_tot / 20, zero_mean = uplink ? zero_tot / 20 : zero_tot / 16;
    SyncCheck r;
    r.center       = (int)(int16_t)((one_mean + zero_mean) / 2);
    const bool bad = (one && d < r.center) || (zero && d > r.center);
    r.ok           = __builtin_popcountll(__ballot(bad)) <= 4;
    return r;
}
__device__ __forceinline__ uint64_t sign_window_tile(const int16_t* dphi_s, int off, int lane)
{
    return __ballot(dphi_s[off + 2 * (lane & 31) + (lane >> 5)] > 0);
}
// frame bytes b0 .. b1 - 1 of a frame whose bit 0 sits at tile index off0, into out[]: lane l slices byte b0 + l from the eight
// differences two samples apart -- packed in pairs, centre - d saturated (negative exactly when d > centre), the sign bits
// gathered MSB-first by v_dot2.  (Round-1 form: 64 bits per ballot, eight lanes unpacking each ballot: 3.5 times the instructions.)
__device__ __forceinline__ void slice_bytes_tile(const int16_t* dphi_s, int off0, int center, int b0, int b1, uint8_t* out, int lane)
{
    typedef unsigned short v2u __attribute__((ext_vector_type(2)));
    for (int bb = b0; bb < b1; bb += 64)
    {
        const int byte = bb + lane;
        if (byte < b1)
        {
            const uint16_t* d  = reinterpret_cast<const uint16_t*>(dphi_s) + (off0 + 16 * byte);
            const uint32_t  c2 = ((uint32_t)center & 0xFFFFu) * 0x00010001u;
            uint32_t        acc = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const uint32_t pair = (uint32_t)d[4 * k] | ((uint32_t)d[4 * k + 2] << 16);
                uint32_t       t;
                asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(t) : "v"(c2), "v"(pair));
                acc = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u, t & 0x80008000u), (v2u){(unsigned short)(128 >> (2 * k)), (unsigned short)(64 >> (2 * k))}, acc, false);
            }
            out[byte] = (uint8_t)(acc >> 15);
        }
    }
}

// Behind an `if (lane == 0) store;` that is followed by a jump (break): keeps the two paths' meeting point a block of its own.  Without it
// the compiler folds that meeting point into the jump's target, the target's phi nodes then have the lane-dependent branch among their
// predecessors, and every value carried through them (all of the demodulating wave's uniform state) counts as lane-varying: vector
// registers and exec-mask branches instead of scalar ones.  No instruction is emitted for it.
__device__ __forceinline__ void lane0_join() { __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"), __builtin_amdgcn_wave_barrier(), __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// ---- Reed-Solomon with the whole wave on one code word.  Same procedure and same results as rs978_decode_with_syndromes
// (rs978.h; the host build of that one is what the CPU tests hold against the oracle, and tests/test_uat978_gpu.py holds
// this one against it on random words), laid out across lanes: lane i owns coefficient i of lambda and b during
// Berlekamp-Massey, the Chien search tries four field elements per lane, Forney runs one lane per root.
// w.s[0 .. nr) holds the syndromes.  Every lane must call; the result is the same in every lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_zero_x(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ int wave_xor_i(int x)
{
    x ^= dpp_or_zero_x<0x111, 0xF>(x);
    x ^= dpp_or_zero_x<0x112, 0xF>(x);
    x ^= dpp_or_zero_x<0x114, 0xF>(x);
    x ^= dpp_or_zero_x<0x118, 0xF>(x);
    x ^= dpp_or_zero_x<0x142, 0xA>(x);
    x ^= dpp_or_zero_x<0x143, 0xC>(x);
    return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ int row0_xor_i(int x)
{ // XOR over lanes 0 .. 15 only
    x ^= dpp_or_zero_x<0x111, 0xF>(x);
    x ^= dpp_or_zero_x<0x112, 0xF>(x);
    x ^= dpp_or_zero_x<0x114, 0xF>(x);
    x ^= dpp_or_zero_x<0x118, 0xF>(x);
    return __builtin_amdgcn_readlane(x, 15);
}
__device__ __forceinline__ int      gf_fold(int x) { return (x & 255) + (x >> 8); } // == x mod 255 as an index into exp[] (< 510 for x < 65536)

// The nr syndromes of the n symbols data[0], data[stride], ... with the wave across the symbols instead of Horner's n dependent
// steps: S_i = sum_j data[j] alpha^((fcr + i)(n - 1 - j)); lane l takes symbols l and l + 64, four syndromes share one XOR
// reduction (one byte of a dword each).  Same values as rs978_syndrome.  out[] is in LDS; the caller fences.
__device__ __forceinline__ int mod255_16(int x)
{ // x < 65536 -> congruent value in [0, 256]
    x = (x & 255) + (x >> 8);
    return (x & 255) + (x >> 8);
}
typedef __attribute__((address_space(3))) const uint8_t* lds_cu8; // kept out of line (registers), so the address space has to be spelled out
typedef __attribute__((address_space(3))) uint8_t*       lds_u8;
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class P>
__device__ __forceinline__ P uni_lds(P p)
{ // a function's arguments arrive in vector registers; what is the same in every lane is said to be so: scalar loop control, scalar bases
    return (P)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p);
}
__device__ __noinline__ void syndromes_lds(lds_cu8 exp_t, lds_cu8 log_t, int nr, int n, lds_cu8 data, int stride, lds_u8 out, int lane)
{
    exp_t = uni_lds(exp_t), log_t = uni_lds(log_t), data = uni_lds(data), out = uni_lds(out), nr = uni(nr), n = uni(n), stride = uni(stride);
    const bool     two = n > 64; // uniform
    const int      j0 = lane, j1 = lane + 64;
    const uint32_t d0 = j0 < n ? data[j0 * stride] : 0u, d1 = (two && j1 < n) ? data[j1 * stride] : 0u;
    const int      l0 = log_t[d0], l1 = log_t[d1];
    const uint32_t p0 = j0 < n ? (uint32_t)(n - 1 - j0) : 0u, p1 = j1 < n ? (uint32_t)(n - 1 - j1) : 0u; // < 255
    // exponent of alpha^(root * p) for root = fcr, then + p per syndrome, kept below 255 (min with the wrapped difference)
    uint32_t e0 = (uint32_t)mod255_16(kRsFcr * (int)p0), e1 = (uint32_t)mod255_16(kRsFcr * (int)p1);
    e0 = __builtin_elementwise_min(e0, e0 - 255u), e1 = __builtin_elementwise_min(e1, e1 - 255u);
    for (int i0 = 0; i0 < nr; i0 += 4)
    {
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            uint32_t term = d0 ? (uint32_t)exp_t[l0 + (int)e0] : 0u;
            e0 += p0, e0 = __builtin_elementwise_min(e0, e0 - 255u);
            if (two)
            {
                term ^= d1 ? (uint32_t)exp_t[l1 + (int)e1] : 0u;
                e1 += p1, e1 = __builtin_elementwise_min(e1, e1 - 255u);
            }
            packed |= term << (8 * k);
        }
        const uint32_t red = (uint32_t)wave_xor_i((int)packed);
        if (lane < 4 && i0 + lane < nr) out[i0 + lane] = (uint8_t)(red >> (8 * lane));
    }
}
__device__ __forceinline__ void syndromes_wave(const RsTables& T, int nr, int n, const uint8_t* data, int stride, uint8_t* out, int lane)
{
    syndromes_lds((lds_cu8)T.exp, (lds_cu8)T.log, nr, n, (lds_cu8)data, stride, (lds_u8)out, lane);
}

// The same for the code that nearly every match asks for first -- the 14 syndromes of 48 bytes (RS(48,34)) -- with the exponents
// (fcr + i)(n - 1 - lane) mod 255 kept in registers for the whole launch (syndrome_exponents; likewise the 12 of 30 bytes): a syndrome
// costs one address add and one table look-up instead of the six instructions of stepping the exponent modulo 255, and the test for a zero
// byte is made once per four syndromes.  Same values as syndromes_lds(14, 48, stride 1) (tests: every stream comparison goes through it).
template <int GROUPS>
struct CodeExponents
{
    uint32_t pk[GROUPS]; // byte i % 4 of pk[i / 4] = (fcr + i)(n - 1 - lane) mod 255
};
template <int N, int GROUPS>
__device__ __forceinline__ CodeExponents<GROUPS> syndrome_exponents(int lane)
{
    CodeExponents<GROUPS> x;
    const uint32_t        p = lane < N ? (uint32_t)(N - 1 - lane) : 0u;
    uint32_t              e = (uint32_t)mod255_16(kRsFcr * (int)p);
    e                       = __builtin_elementwise_min(e, e - 255u);
#pragma unroll
    for (int g = 0; g < GROUPS; g++)
    {
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            w |= e << (8 * k);
            e += p, e = __builtin_elementwise_min(e, e - 255u);
        }
        x.pk[g] = w;
    }
    return x;
}
typedef CodeExponents<4> LongExponents;  // RS(48,34): 14 syndromes
typedef CodeExponents<3> ShortExponents; // RS(30,18): 12 syndromes
template <int N, int NR, int GROUPS>
__device__ __forceinline__ void syndromes_fixed_wave(const RsTables& T, const uint8_t* data, uint8_t* out, int lane, const CodeExponents<GROUPS>& x)
{
    static_assert(4 * GROUPS >= NR && N <= 64, "one byte per lane, four syndromes per register");
    const uint32_t d  = lane < N ? (uint32_t)data[lane] : 0u;
    const uint8_t* eb = T.exp + T.log[d]; // (log[0] = 0: a zero byte's terms are dropped below)
#pragma unroll
    for (int g = 0; g < GROUPS; g++)
    {
        const uint32_t w      = x.pk[g];
        uint32_t       packed = (uint32_t)eb[w & 255u] | (uint32_t)eb[(w >> 8) & 255u] << 8 | (uint32_t)eb[(w >> 16) & 255u] << 16 | (uint32_t)eb[w >> 24] << 24;
        packed                = d ? packed : 0u;
        const uint32_t red    = (uint32_t)wave_xor_i((int)packed);
        if (lane < 4 && 4 * g + lane < NR) out[4 * g + lane] = (uint8_t)(red >> (8 * lane));
    }
}

// ---- The error locator without Berlekamp-Massey (round 5), for the two ADS-B codes (t = nr / 2 = 7 or 6 fits a wave: t rows of t + 1 entries).
// Berlekamp-Massey is nr dependent iterations of ~33 vector instructions and three LDS round trips each, with the wave serving 15 lanes' worth of
// coefficients: a third of the demodulation kernel, most of it spent on words that are NOT code words of the code tried (a short frame goes
// through the long code first, correct_adsb_frame's order) and need the full nr iterations to say so.  What it computes is the shortest linear
// recurrence s_k = sum_{j=1..L} lambda_j s_{k-j} generating the nr syndromes.  Whenever 2 L <= nr that recurrence is unique, so ANY way of finding
// it gives Berlekamp-Massey's lambda -- here Gauss-Jordan elimination, without row exchanges, on the t equations k = t+1 .. 2t in the t unknowns
// lambda_1 .. lambda_t (entry (i, j) = s_{t+i-j}, right-hand side s_{t+i+1}; lane 8 i + j holds one entry, value and logarithm), all rows at once:
//   * t pivots: the matrix is regular, L = t, lambda is the solution (a word outside the code's reach -- the common case);
//   * the pivot of step r vanishes and every row from r on has become zero: the system has rank r with the first r columns as pivot columns,
//     lambda_1 .. lambda_r from the right-hand sides (the others zero) satisfy the equations k = t+1 .. 2t; if they also satisfy k = r+1 .. t
//     (checked) the recurrence has length r, and a shorter one would have made the rank smaller: L = r (a word with r <= t errors);
//   * anything else (a vanishing pivot with non-zero rows left: a leading minor is singular, or L > t): -1, and the caller runs Berlekamp-Massey
//     as before -- a few words in a hundred.
// Seven steps of ~14 vector instructions and two LDS-crossbar fetches instead of fourteen of 33.  Held against the sequential decoder
// (rs978.h, itself held against the oracle) on 12 000 clean, correctable, uncorrectable and random words of the two codes
// (tests/test_uat978_gpu.py::test_device_reed_solomon_matches_oracle_including_beyond_capacity) and by every stream comparison.
// syn / lsyn: lane l < nr holds syndrome l and its logarithm.  Returns L (0 .. t) with *lam_out = coefficient `lane` of lambda (lane 0: 1), or -1.
template <int T, class Tables>
__device__ __forceinline__ int locator_by_elimination(const Tables& tb, uint32_t syn, uint32_t lsyn, int lane, uint32_t* lam_out)
{
    static_assert(T + 1 <= 8 && 8 * T <= 64, "rows of eight lanes");
    const int      row = lane >> 3, col = lane & 7;
    const bool     in  = row < T && col <= T;
    const uint32_t pk  = syn | (lsyn << 8); // (value, logarithm) of the lane's syndrome
    auto fetch = [](int from_lane, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(4 * from_lane, (int)v); };
    uint32_t e = fetch(col < T ? T + row - col - 1 : T + row, pk); // s (0-based index): entry (row, col) = s_{t+row-col} (1-based), rhs s_{t+row+1}
    e          = in ? e : 0u;
    uint32_t v = e & 255u, lv = e >> 8;
    int      rank = T;
#pragma unroll
    for (int p = 0; p < T; p++)
    {
        const uint32_t piv = (uint32_t)__builtin_amdgcn_readlane((int)(v | (lv << 8)), 9 * p);
        if ((piv & 255u) == 0u)
        {
            rank = p;
            break;
        }
        // row p divided by its pivot
        uint32_t nlv = lv + 255u - (piv >> 8);
        nlv          = nlv >= 255u ? nlv - 255u : nlv;
        if (row == p) lv = nlv, v = v ? (uint32_t)tb.exp[nlv] : 0u;
        // every other row: entry (i, j) += entry (i, p) * new entry (p, j)
        const uint32_t pe = v | (lv << 8);
        const uint32_t rp = fetch(8 * p + col, pe), f = fetch(8 * row + p, pe);
        if (row != p && in)
        {
            v ^= ((f & 255u) && (rp & 255u)) ? (uint32_t)tb.exp[(f >> 8) + (rp >> 8)] : 0u;
            lv = tb.log[v];
        }
    }
    if (__ballot(in && row >= rank && v != 0u) != 0) return -1; // rows left over: no recurrence of length <= t from these equations
    // lambda_j = right-hand side of row j - 1
    const uint32_t sol = fetch(8 * ((lane - 1) & 7) + T, v | (lv << 8));
    const uint32_t lam = lane == 0 ? 1u : (lane <= rank ? (sol & 255u) : 0u);
    if (rank < T)
    { // the equations k = rank+1 .. t (1-based) are not among those solved: lane m checks k = rank + 1 + m
        const uint32_t llam = lane >= 1 && lane <= rank ? sol >> 8 : 0u;
        const int      k    = rank + 1 + lane; // 1-based index of the syndrome on the left
        uint32_t       acc  = 0;
#pragma unroll
        for (int j = 1; j < T; j++)
        {
            const uint32_t lj = (uint32_t)__builtin_amdgcn_readlane((int)(lam | (llam << 8)), j); // lambda_j (uniform)
            const int      from = k - j - 1;                                                       // s_{k-j}, 0-based
            const uint32_t sj   = fetch(from >= 0 && from < 2 * T ? from : 0, pk);
            if (j <= rank && (lj & 255u) && (sj & 255u)) acc ^= (uint32_t)tb.exp[(lj >> 8) + (sj >> 8)];
        }
        const uint32_t left = fetch(k - 1 < 2 * T ? k - 1 : 0, pk) & 255u;
        if (__ballot(lane < T - rank && acc != left) != 0) return -1;
    }
    *lam_out = lam;
    return rank;
}

// Kept out of line (three call sites: long and short ADS-B code, uplink blocks; inlined copies cost the kernel a wave of occupancy), so
// the LDS address space of its operands is spelled out as for syndromes_lds.
typedef __attribute__((address_space(3))) RsWork* lds_work;
__device__ __noinline__ int rs_decode_lds(lds_cu8 exp_t, lds_cu8 log_t, int nr, int pad, lds_u8 data, int stride, lds_work wp, int lane)
{
    exp_t = uni_lds(exp_t), log_t = uni_lds(log_t), data = uni_lds(data), wp = uni_lds(wp), nr = uni(nr), pad = uni(pad), stride = uni(stride);
    auto& w = *wp;
    struct
    {
        lds_cu8 exp, log;
    } T = {exp_t, log_t};
    auto gf_mul = [&](uint32_t a, uint32_t b) -> uint32_t { return (a && b) ? T.exp[T.log[a] + T.log[b]] : 0u; };
    const uint32_t syn = lane < nr ? w.s[lane] : 0u;
    if (__ballot(syn != 0) == 0) return 0;

    // Berlekamp-Massey: lane i holds lambda[i] and b[i].  Kept beside their logarithms so that an iteration costs four table look-ups
    // instead of nine: b is only ever multiplied (its value is needed as "zero or not" + logarithm), the syndrome window
    // s[r-1-i] of lane i is last iteration's window of lane i-1 (a DPP shift of the value and of its logarithm, the new
    // syndrome entering at lane 0), and b = lambda / discrepancy is a subtraction of logarithms.
    const uint32_t lsyn = T.log[syn];
    uint32_t       lam = lane == 0, llam = 0; // log[1] = 0
    // the locator by elimination where the code is small enough for it (the two ADS-B codes), by Berlekamp-Massey otherwise or when that declines
    int fast = -1;
    if (nr == 14) fast = locator_by_elimination<7>(T, syn, lsyn, lane, &lam);
    else if (nr == 12) fast = locator_by_elimination<6>(T, syn, lsyn, lane, &lam);
    uint32_t       bnz = lane == 0, lbb = 0;
    uint32_t       sv = 0, lsv = 0;
    int            el = 0;
    if (fast < 0) lam = lane == 0;
    for (int r = 1; fast < 0 && r <= nr; r++)
    {
        const uint32_t s_new = (uint32_t)__builtin_amdgcn_readlane((int)syn, r - 1), ls_new = (uint32_t)__builtin_amdgcn_readlane((int)lsyn, r - 1);
        sv  = (uint32_t)__builtin_amdgcn_update_dpp((int)s_new, (int)sv, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        lsv = (uint32_t)__builtin_amdgcn_update_dpp((int)ls_new, (int)lsv, 0x138, 0xF, 0xF, false);
        const uint32_t term   = (lam && sv) ? (uint32_t)T.exp[llam + lsv] : 0u;
        const uint32_t discr  = (uint32_t)(nr <= 15 ? row0_xor_i((int)term) : wave_xor_i((int)term)); // nr + 1 coefficients: lanes above hold zero
        const uint32_t bs_nz  = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bnz, 0x138, 0xF, 0xF, false);
        const uint32_t lbs    = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lbb, 0x138, 0xF, 0xF, false);
        if (discr == 0) bnz = bs_nz, lbb = lbs;
        else
        {
            const uint32_t ldis = T.log[discr];
            const uint32_t t    = lam ^ (bs_nz ? (uint32_t)T.exp[ldis + lbs] : 0u);
            if (2 * el <= r - 1)
            { // b = lambda / discr
                el             = r - el;
                const uint32_t q = llam + 255u - ldis; // 1 .. 509
                bnz = lam != 0, lbb = q >= 255u ? q - 255u : q;
            }
            else bnz = bs_nz, lbb = lbs;
            lam = t, llam = T.log[t];
        }
        if (lane > nr) lam = 0, bnz = 0;
    }
    const uint64_t nz  = __ballot(lam != 0);
    const int      deg = nz ? 63 - __builtin_clzll(nz) : 0;
    if (lane <= nr) w.lambda[lane] = (uint8_t)lam;
    wave_fence();

    // Chien search: X^-1 = alpha^i for i = 1 .. 255, four per lane; the coefficient and its logarithm are fetched once for the four.
    // (Round 6 measured the field elements of the code word's own positions first -- a lane's fourth element for the two ADS-B words, third and fourth for
    // an uplink block; deg roots there are all there are -- and the rest only when roots are missing: the kernel went from 0.2008 to 0.2125 ms.  The
    // words that gain are the correctable ones with errors, a quarter of the frames; every short frame's failing long-code attempt, 29 % of them,
    // evaluates in two passes instead of one.  profiles/r06_uat978_variants.txt)
    int      count = 0;
    uint32_t q4[4] = {1, 1, 1, 1};
    for (int j = 1; j <= deg; j++)
    {
        const uint32_t lj = w.lambda[j]; // uniform
        if (lj == 0) continue;
        const int ll = T.log[lj], step = j * (1 + lane);
#pragma unroll
        for (int k = 0; k < 4; k++) q4[k] ^= T.exp[gf_fold(ll + step + 64 * j * k)];
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const int      i    = 1 + lane + 64 * k;
        const bool     hit  = i <= 255 && q4[k] == 0;
        const uint64_t hits = __ballot(hit);
        if (hit)
        {
            const int at = count + __builtin_popcountll(hits & ((1ull << lane) - 1ull));
            if (at < kRsMaxRoots) w.root[at] = (uint8_t)i, w.loc[at] = (uint8_t)(i - 1);
        }
        count += __builtin_popcountll(hits);
    }
    if (count != deg) return -1;

    // omega(x) = s(x) lambda(x) mod x^deg, one coefficient per lane
    if (lane < deg)
    {
        uint32_t acc = 0;
        for (int j = 0; j <= lane; j++) acc ^= gf_mul(w.s[lane - j], w.lambda[j]);
        w.omega[lane] = (uint8_t)acc;
    }
    wave_fence();
    // Forney, one root per lane
    if (lane < count)
    {
        const int rt   = w.root[lane];
        uint32_t  num1 = 0, den = 0;
        for (int i = deg - 1; i >= 0; i--)
            if (w.omega[i]) num1 ^= T.exp[gf_fold(T.log[w.omega[i]] + i * rt)];
        const uint32_t num2 = T.exp[gf_fold(rt * (kRsFcr - 1) + 255)];
        const int      top  = (deg < nr - 1 ? deg : nr - 1) & ~1;
        for (int i = top; i >= 0; i -= 2)
            if (w.lambda[i + 1]) den ^= T.exp[gf_fold(T.log[w.lambda[i + 1]] + i * rt)];
        if (num1 != 0 && (int)w.loc[lane] >= pad)
        {
            const int lden = den ? T.log[den] : 255;
            data[((int)w.loc[lane] - pad) * stride] ^= T.exp[gf_fold(T.log[num1] + T.log[num2] + 255 - lden)];
        }
    }
    wave_fence();
    return count;
}

__device__ __forceinline__ int rs_decode_wave(const RsTables& T, int nr, int pad, uint8_t* data, int stride, RsWork& w, int lane)
{
    // (the same in every lane, but a function's result counts as lane-varying: said here, the caller's loop control stays on the scalar unit)
    return __builtin_amdgcn_readfirstlane(rs_decode_lds((lds_cu8)T.exp, (lds_cu8)T.log, nr, pad, (lds_u8)data, stride, (lds_work)&w, lane));
}

// correct_adsb_frame with the wave on one slicing: w.s = the 14 long syndromes.  Returns the bits to jump (0 = neither); *rs = corrected symbols (9999 = neither).  Uniform.
__device__ __forceinline__ int correct_adsb_wave(const RsTables& T, uint8_t* frame48, RsWork& w, int lane, int* rs, const ShortExponents* short_exp = nullptr)
{
    int n = rs_decode_wave(T, 14, 207, frame48, 1, w, lane);
    if (n >= 0 && n <= 7 && __builtin_amdgcn_readfirstlane(frame48[0] >> 3) != 0)
    {
        *rs = n;
        return kUatLongSkip;
    }
    if (short_exp) syndromes_fixed_wave<30, 12>(T, frame48, w.s, lane, *short_exp); // only now: most frames are long and never get here
    else syndromes_wave(T, 12, 30, frame48, 1, w.s, lane);
    wave_fence();
    n = rs_decode_wave(T, 12, 225, frame48, 1, w, lane);
    if (n >= 0 && n <= 6 && __builtin_amdgcn_readfirstlane(frame48[0] >> 3) == 0)
    {
        *rs = n;
        return kUatShortSkip;
    }
    *rs = 9999;
    return 0;
}

// test hook: decode `count` independent code words of one code, one wave each (kind 0/1/2 as in adsb_amd_uat_rs_decode)
__global__ __launch_bounds__(64) void uat_rs_selftest_kernel(const RsTables* __restrict__ rs_tables, int kind, uint8_t* words, int* results, int count)
{
    __shared__ RsTables T;
    __shared__ RsWork   w;
    __shared__ uint8_t  cw[96];
    const int lane = threadIdx.x;
    const int nr = kind == 0 ? 12 : kind == 1 ? 14 : 20, pad = kind == 0 ? 225 : kind == 1 ? 207 : 163, n = 255 - pad;
    for (int i = lane; i < (int)sizeof(RsTables) / 4; i += 64) reinterpret_cast<uint32_t*>(&T)[i] = reinterpret_cast<const uint32_t*>(rs_tables)[i];
    wave_fence();
    for (int c = blockIdx.x; c < count; c += gridDim.x)
    {
        for (int i = lane; i < n; i += 64) cw[i] = words[(size_t)c * n + i];
        wave_fence();
        if (lane < nr) w.s[lane] = rs978_syndrome(T, lane, n, cw, 1);
        wave_fence();
        const int r = rs_decode_wave(T, nr, pad, cw, 1, w, lane);
        for (int i = lane; i < n; i += 64) words[(size_t)c * n + i] = cw[i];
        if (lane == 0) results[c] = r;
        wave_fence();
    }
}

// ---- the scan loop after a jump.  Taking a frame moves the loop `skip` bits ahead without clearing its two 18-bit shift registers
// (one per sample alignment), so for the next 17 bits each holds (18 - t) bits from before the jump and t new ones and can fire where
// the stream itself has no match.  Registers here are in stream order (bit k = the k-th oldest bit, as uat_rec_t::window delivers
// them); the check words are the first 18 bits of the sync words in that order.
constexpr uint32_t kCheckMask = (1u << kUatCheckBits) - 1u;
constexpr uint32_t stream_order18(uint64_t sync36)
{
    uint32_t r = 0;
    for (int k = 0; k < 18; k++) r |= (uint32_t)((sync36 >> (35 - k)) & 1u) << k;
    return r;
}
constexpr uint32_t kCheckT0 = stream_order18(kAdsbSync), kCheckT1 = stream_order18(kUplinkSync);

struct StaleWindow
{
    int64_t  bit = 0;          // first bit examined after the jump
    uint32_t steps = 0;        // bit t: step t (1 .. 17, examining bit `bit + t - 1`) fires and has not been tried yet (wave-uniform)
    uint32_t w0_step = 0, w1_step = 0;   // lane t: both registers at step t
    uint32_t w0_fired = 0, w1_fired = 0; // the registers at the step taken last (wave-uniform)

    // old0 / old1: the registers at the jump; fresh: the sign bits that enter them (low word even samples, high word odd)
    __device__ __forceinline__ void jump(uint32_t old0, uint32_t old1, uint64_t fresh, int64_t first_bit, int64_t lenbits, int lane)
    {
        bit             = first_bit;
        const bool in   = lane >= 1 && lane <= 17 && first_bit + lane - 1 < lenbits;
        const int  t    = in ? lane : 1;
        w0_step         = ((old0 >> t) | ((uint32_t)fresh << (18 - t))) & kCheckMask;
        w1_step         = ((old1 >> t) | ((uint32_t)(fresh >> 32) << (18 - t))) & kCheckMask;
        const bool fire = in && (w0_step == kCheckT0 || w1_step == kCheckT0 || w0_step == kCheckT1 || w1_step == kCheckT1);
        steps           = (uint32_t)__ballot(fire);
    }
};

// Measurement builds (diag.hip.h, DIAG_UAT): shader-clock cycles of the demodulating wave by phase and kind of match, summed over a launch
// (tools/uat_diag.py).  Every macro is empty in the product build.
#if DIAG_UAT
__device__ unsigned long long g_uat_diag[2][8]; // [kind][phase]; phase 7 = positions demodulated
#define UAT_DIAG_DECLARE() unsigned long long diag_acc[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}}, diag_t = 0
#define UAT_DIAG_BEGIN() diag_t = __builtin_readcyclecounter()
#define UAT_DIAG_LAP(phase)                                             \
    do                                                                  \
    {                                                                   \
        const unsigned long long diag_n = __builtin_readcyclecounter(); \
        diag_acc[kind][phase] += diag_n - diag_t, diag_t = diag_n;      \
    } while (0)
#define UAT_DIAG_END(kind) diag_acc[kind][7] += 1
#define UAT_DIAG_FLUSH()                                                                           \
    do                                                                                             \
    { /* once per wave, after its last position */                                                 \
        if (lane == 0)                                                                             \
            for (int diag_k = 0; diag_k < 2; diag_k++)                                             \
                for (int diag_p = 0; diag_p < 8; diag_p++)                                         \
                    if (diag_acc[diag_k][diag_p]) atomicAdd(&g_uat_diag[diag_k][diag_p], diag_acc[diag_k][diag_p]); \
    } while (0)
#else
#define UAT_DIAG_DECLARE() (void)0
#define UAT_DIAG_BEGIN() (void)0
#define UAT_DIAG_LAP(phase) (void)0
#define UAT_DIAG_END(kind) (void)0
#define UAT_DIAG_FLUSH() (void)0
#endif
enum { kDiagStage = 0, kDiagSync, kDiagSlice, kDiagSyndromes, kDiagDecode, kDiagMoreTiles, kDiagOutput };

// Waves of a demodulating workgroup: they share the folded phase table (32 KB) and the Reed-Solomon tables (768 B) and have 4 032 B each of their own.
// Two workgroups per CU: 2 x (32 768 + 768 + 12 x 4 032) = 163 840 bytes, ALL of a CU's LDS (the static_assert in the kernel holds it there: one byte
// more and a CU holds one workgroup, half the waves, without any error).  Nothing else that needs LDS runs on a CU while two of these are resident.
constexpr int kUatDemodWaves = 12;
constexpr size_t kCuLdsBytes = 160 * 1024;
template <bool PHASES_GIVEN>
__global__ __launch_bounds__(64 * kUatDemodWaves, 6) void uat_demod_kernel(const uint16_t* __restrict__ in, const uint16_t* __restrict__ lut, uint64_t n,
                                                       const RsTables* __restrict__ rs_tables, const uint32_t* __restrict__ cand, uint32_t ncand,
                                                       uat_rec_t* __restrict__ recs, uat_win_t* __restrict__ wins, uint8_t* __restrict__ payloads, uint8_t* __restrict__ uplink_payloads,
                                                       uint32_t uplink_cap,
                                                       uint32_t* __restrict__ uplink_count, uint32_t* __restrict__ work_counters, uint32_t nranges,
                                                       const uint32_t* __restrict__ up_list, const uint32_t* __restrict__ up_count, uint32_t single_word,
                                                       int64_t lenbits, uint32_t* __restrict__ next_bit, uat_extra_t* __restrict__ extras,
                                                       uint8_t* __restrict__ extra_payloads, uint32_t extra_cap, uint32_t* __restrict__ counts)
{
    struct WaveArea // a wave's own
    {
        __attribute__((aligned(16))) int16_t dphi_s[kUatTile];
        uint8_t raw[2][kUatUplinkBytes + 8];
        RsWork  work[6];
    };
    __shared__ RsTables T;
    __shared__ __attribute__((aligned(16))) uint16_t folded[PHASES_GIVEN ? 2 : kUatFoldedEntries];
    __shared__ WaveArea areas[kUatDemodWaves];
    static_assert(2 * (sizeof(RsTables) + sizeof(uint16_t) * kUatFoldedEntries + sizeof(WaveArea) * kUatDemodWaves) <= kCuLdsBytes, "two demodulating workgroups per CU");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (said to be uniform: the wave's LDS addresses stay on the scalar side)
    for (int i = threadIdx.x; i < (int)sizeof(RsTables) / 4; i += 64 * kUatDemodWaves) reinterpret_cast<uint32_t*>(&T)[i] = reinterpret_cast<const uint32_t*>(rs_tables)[i];
    if constexpr (!PHASES_GIVEN)
        for (int i = threadIdx.x; i < kUatFoldedEntries / 2; i += 64 * kUatDemodWaves)
        { // entries (mq, mi) and (mq, mi + 1) = lut[128 + mq][128 + mi ..]: one dword of the table's row 128 + mq
            const int mq = i >> 6, mi = 2 * (i & 63);
            reinterpret_cast<uint32_t*>(folded)[i ^ (mq & 63)] = *reinterpret_cast<const uint32_t*>(lut + (((128 + mq) << 8) | (128 + mi))); // (lut2_folded's bank swizzle)
        }
    __syncthreads(); // the only one: from here on the waves go their own ways
    auto& raw    = areas[wave].raw;
    auto& work   = areas[wave].work;
    auto& dphi_s = areas[wave].dphi_s;

    // Candidates differ a lot in cost (noise that fails the sync re-check, short frames whose long-code attempt has to fail
    // first, uplink frames with twelve code words), so they are handed out by work counters, not by a fixed stride: the list
    // is cut into `nranges` pieces, each with a counter on its own cache line; a block takes its first candidate by position.
    // (wave w of the grid's gridDim.x * kUatDemodWaves: range w mod nranges, slot w / nranges; the waves beyond nranges * nslot have nothing to do)
    const uint32_t gwave = blockIdx.x * kUatDemodWaves + (uint32_t)wave, nslot = gridDim.x * kUatDemodWaves / nranges;
    const uint32_t range = gwave % nranges, slot = gwave / nranges;
    const uint32_t per   = (ncand + nranges - 1) / nranges;
    const uint32_t first = range * per < ncand ? range * per : ncand, end = first + per < ncand ? first + per : ncand;
    // Uplink matches first (up_list, dealt round-robin to the ranges), then the range's own slice of the list with the uplink
    // entries passed over.  up_list == nullptr (single look-ups): plain order.
    const uint32_t nup    = up_list ? *up_count : 0u;
    const uint32_t nup_r  = nup > range ? (nup - range + nranges - 1) / nranges : 0u;
    const uint32_t nitems = nup_r + (end - first);
    const g_u32    my_counter = (g_u32)&work_counters[range * 32u];
    const LongExponents  long_exp  = syndrome_exponents<48, 4>(lane);
    const ShortExponents short_exp = syndrome_exponents<30, 3>(lane);
    UAT_DIAG_DECLARE();
    for (uint32_t item = slot < nslot ? slot : nitems; item < nitems;)
    {
        const bool     from_list = item < nup_r;
        const uint32_t c         = from_list ? up_list[range + item * nranges] : first + (item - nup_r);
        uint32_t   word = cand ? cand[c] : single_word; // cand == nullptr: one look-up the host asked for, passed by value
        uat_rec_t* r    = &recs[c];
        uat_win_t* wn   = &wins[c]; // (null behind a frame: the frames the loop reaches through stale register bits carry no windows of their own)
        uint8_t*   pay  = payloads + (size_t)c * kUatPayloadStride;
        if (!from_list && (word >> 31) && up_list)
        { // an uplink match inside the slice: taken in the first phase
            item = nslot + draw_ticket(my_counter, lane);
            continue;
        }
        // The ticket for the wave's next match is drawn inside the staging of this one's first tile (stage_dphi): the atomic's round trip
        // to the memory side runs beside the tile's loads instead of being a wait of its own at the head of every match.  (That staging
        // stands in front of the loop over positions so that the ticket is a value of THIS loop: the inner one's exits count as
        // lane-varying, and a value carried through it would live in a vector register.)
        wave_fence(); // the previous match's readers are done with the tile
        const uint32_t next_item = nslot + (uint32_t)__builtin_amdgcn_readfirstlane((int)stage_dphi<PHASES_GIVEN>(
                                               (g_cu16)in, (g_cu16)lut, (lds_cu16)folded, n, (uint64_t)(word & 0x7FFFFFF8u), (lds_i16)dphi_s, lane, my_counter));
        // After the match's own frame: the frames the scan loop would take behind it through stale register bits (see StaleWindow).
        // They are demodulated by this wave, by the same code: the body below runs once per position.  All of this is wave-uniform.
        StaleWindow stale;
        bool        chained = false; // the position being demodulated is such an extra, not the match
        uint32_t    seq     = 0;
        uint32_t    sink    = 0; // measurement builds (diag::kUatParts): what keeps the parts left in from being optimised away
        for (;;)
        {
        const uint32_t kind = word >> 31;
        bool cut = false; // (a match that is cut short leaves as one at which nothing decodes)
        if constexpr (diag::kUatParts <= 6) cut = kind != 0;
        if constexpr (diag::kUatParts == 1) cut = true;
        const uint64_t idx  = word & 0x7FFFFFFFu;
        const uint64_t sb   = idx >> 1;
        const uint64_t base = (2 * sb) & ~7ull;           // tile 0 starts here (16-byte aligned in the stream)
        const int      o    = (int)(idx - base);          // the match's first sync sample inside tile 0 (0 .. 8)
        const int      oe   = (int)(2 * sb - base);       // the same, forced even: what the two shift registers are aligned to
        UAT_DIAG_BEGIN();
        if (chained)
        {
            wave_fence(); // the previous position's readers are done with the tile
            stage_dphi<PHASES_GIVEN>((g_cu16)in, (g_cu16)lut, (lds_cu16)folded, n, base, (lds_i16)dphi_s, lane, (g_u32) nullptr);
        }
        wave_fence();
        UAT_DIAG_LAP(kDiagStage);
        const uint64_t w0 = sign_window_tile(dphi_s, oe, lane);
        uint64_t       w1 = 0, w2 = 0;
        const int      nbits = kind ? kUatUplinkBits : kUatLongBytes * 8;
        // frames from sample idx (variant 0) and idx + 1 (variant 1); all of this is wave-uniform.  The variant with fewer corrected
        // symbols is taken and the first on a tie, so a variant 0 that decodes without corrections makes variant 1 irrelevant.
        int skip0 = 0, skip1 = 0, rs0 = 9999, rs1 = 9999;
        if (cut) sink += (uint32_t)dphi_s[word & 63u];
        else if (kind == 0)
        { // everything an ADS-B match needs lies in tile 0 (the last window ends 914 samples after its start)
            w1 = sign_window_tile(dphi_s, oe + 2 * (kUatShortSkip + 1), lane);
            w2 = sign_window_tile(dphi_s, oe + 2 * (kUatLongSkip + 1), lane);
            bool sliced0 = false; // sliced0: sliced[] holds variant 0's 48 bytes as sliced, before any correction
            static_assert(sizeof(RsWork) >= kUatLongBytes, "room for the copy");
            uint8_t* const sliced = reinterpret_cast<uint8_t*>(&work[2]); // (an ADS-B match works in work[0] and work[1] only)
#pragma unroll 1
            for (int v = 0; v < 2; v++)
            {
                if (v == 1 && skip0 && rs0 == 0) break;
                const SyncCheck sc = check_sync_tile(dphi_s, o + v, false, lane);
                UAT_DIAG_LAP(kDiagSync);
                if constexpr (diag::kUatParts == 2)
                {
                    sink += (uint32_t)sc.ok + (uint32_t)sc.center;
                    continue;
                }
                if (!sc.ok) continue;
                slice_bytes_tile(dphi_s, o + v + 72, sc.center, 0, nbits / 8, raw[v], lane);
                wave_fence();
                UAT_DIAG_LAP(kDiagSlice);
                if constexpr (diag::kUatParts == 3)
                {
                    sink += raw[v][word & 31u];
                    continue;
                }
                if (v == 1 && sliced0 && __ballot(lane < kUatLongBytes && raw[1][lane] != sliced[lane]) == 0)
                { // Both alignments slice to the same 48 bytes (two samples per bit: the rule for a signal well above the noise), so
                  // they decode alike and the tie goes to variant 0: nothing left to do for variant 1.
                    skip1 = skip0, rs1 = rs0;
                    break;
                }
                syndromes_fixed_wave<48, 14>(T, raw[v], work[v].s, lane, long_exp);
                wave_fence();
                UAT_DIAG_LAP(kDiagSyndromes);
                if constexpr (diag::kUatParts == 4)
                {
                    sink += work[v].s[word & 7u];
                    if (v == 0)
                    { // (as below, so that the second alignment is passed over as often as in the product)
                        if (lane < kUatLongBytes) sliced[lane] = raw[0][lane];
                        lane0_join();
                        sliced0 = true;
                    }
                    continue;
                }
                if (v == 0)
                { // the decoders correct in place
                    if (lane < kUatLongBytes) sliced[lane] = raw[0][lane];
                    lane0_join();
                    sliced0 = true;
                }
                int       rs_v   = 9999;
                const int skip_v = __builtin_amdgcn_readfirstlane(correct_adsb_wave(T, raw[v], work[v], lane, &rs_v, &short_exp));
                rs_v             = __builtin_amdgcn_readfirstlane(rs_v);
                UAT_DIAG_LAP(kDiagDecode);
                if constexpr (diag::kUatParts == 5)
                {
                    sink += (uint32_t)skip_v + (uint32_t)rs_v;
                    if (v == 0 && skip_v && rs_v == 0) break; // (the product's exit at the head of the loop)
                    continue;
                }
                if (v == 0) skip0 = skip_v, rs0 = rs_v;
                else skip1 = skip_v, rs1 = rs_v;
            }
        }
        else
        {
            bool ok[2];
            int  center[2];
#pragma unroll
            for (int v = 0; v < 2; v++)
            {
                const SyncCheck sc = check_sync_tile(dphi_s, o + v, true, lane);
                ok[v] = sc.ok, center[v] = sc.center;
            }
            UAT_DIAG_LAP(kDiagSync);
            if (ok[0] || ok[1])
            { // an uplink frame spans ten tiles; group g of variant v starts at sample o + v + 72 + 128 g after the first tile's start
                constexpr int kTiles = 10;
                static_assert(72 + 9 + 128 * (kUatUplinkBits / 64 - 1) + 127 < kTiles * kUatTileStride + (kUatTileValid - kUatTileStride), "tiles cover the frame");
                // A tile's samples set out while the tile before it is sliced (nine trips to memory that used to stand between the slicings, one
                // after the other).  The loads are issued without a branch around them -- a value defined on one side of a branch is merged where
                // the sides meet, and the merge waits for it -- so a tile that does not lie wholly inside the stream is loaded from addresses that
                // do (clamped; the stream holds this match: at least 36 samples), its registers are ignored and stage_dphi's guarded path takes it.
                const bool     aligned_in = (reinterpret_cast<uintptr_t>(in) & 15u) == 0;
                const g_cu16   in_16      = (g_cu16)(reinterpret_cast<uintptr_t>(in) & ~(uintptr_t)15u);
                const uint64_t last_oct   = (n & ~7ull) - 8;
                u32x4          nx0 = {0, 0, 0, 0}, nx1 = {0, 0, 0, 0};
                uint32_t       nafter = 0;
                for (int t = 0; t < kTiles; t++)
                {
                    const uint64_t tbase = base + (uint64_t)(t * kUatTileStride);
                    if (t > 0)
                    {
                        wave_fence();
                        if (aligned_in && tbase + (uint64_t)kUatTile + 1 <= n) stage_loaded_tile<PHASES_GIVEN>(nx0, nx1, nafter, (lds_cu16)folded, (lds_i16)dphi_s, lane);
                        else stage_dphi<PHASES_GIVEN>((g_cu16)in, (g_cu16)lut, (lds_cu16)folded, n, tbase, (lds_i16)dphi_s, lane, (g_u32) nullptr);
                        wave_fence();
                        UAT_DIAG_LAP(kDiagMoreTiles);
                    }
                    { // (behind the last tile too: a branch here would be the merge the comment above speaks of)
                        const uint64_t nb = tbase + (uint64_t)kUatTileStride, a0 = nb + 8ull * (uint64_t)lane, a1 = nb + 8ull * (uint64_t)(lane + 64), aa = nb + (uint64_t)kUatTile;
                        nx0 = *reinterpret_cast<g_cu4>(in_16 + (a0 < last_oct ? a0 : last_oct)), nx1 = *reinterpret_cast<g_cu4>(in_16 + (a1 < last_oct ? a1 : last_oct));
                        nafter = in_16[aa < n ? aa : n - 1];
                    }
#pragma unroll
                    for (int v = 0; v < 2; v++)
                        if (ok[v])
                        {
                            const int first = o + v + 72; // of group 0, relative to tile 0
                            int       g0 = (t * kUatTileStride - first + 127) / 128, g1 = ((t + 1) * kUatTileStride - first + 127) / 128;
                            if (t == 0) g0 = 0;
                            if (g1 > nbits / 64) g1 = nbits / 64;
                            slice_bytes_tile(dphi_s, first - t * kUatTileStride, center[v], 8 * g0, 8 * g1, raw[v], lane);
                        }
                    const int after = oe + 2 * (kUatUplinkSkip + 1) - t * kUatTileStride; // the window behind the frame
                    if (after >= 0 && after + 64 < kUatTileValid && after < kUatTileStride) w1 = sign_window_tile(dphi_s, after, lane);
                    UAT_DIAG_LAP(kDiagSlice);
                }
                wave_fence();
                // both alignments sliced to the same 552 bytes: they decode alike, the tie goes to variant 0 (as for ADS-B above)
                bool same = ok[0] && ok[1];
#pragma unroll 1
                for (int k0 = 0; same && k0 < kUatUplinkBytes; k0 += 64) // (a scalar counter: a loop left on a lane's own counter makes `same` lane-varying for the compiler)
                    same = __ballot(k0 + lane < kUatUplinkBytes && raw[0][k0 + lane] != raw[1][k0 + lane]) == 0;
#pragma unroll 1
                for (int v = 0; v < 2; v++)
                {
                    if (v == 1 && same)
                    {
                        skip1 = skip0, rs1 = rs0;
                        continue;
                    }
                    if (!ok[v] || (v == 1 && skip0 && rs0 == 0)) continue;
#pragma unroll 1
                    for (int blk = 0; blk < 6; blk++) syndromes_wave(T, 20, 92, raw[v] + blk, 6, work[blk].s, lane);
                    wave_fence();
                    UAT_DIAG_LAP(kDiagSyndromes);
                    // correct_uplink_frame: every block within 10 corrections
                    int  total = 0;
                    bool good  = true;
#pragma unroll 1
                    for (int blk = 0; blk < 6 && good; blk++)
                    {
                        const int nb = rs_decode_wave(T, 20, 163, raw[v] + blk, 6, work[blk], lane);
                        good         = nb >= 0 && nb <= 10;
                        total += nb;
                    }
                    if (good)
                    {
                        if (v == 0) skip0 = kUatUplinkSkip, rs0 = total;
                        else skip1 = kUatUplinkSkip, rs1 = total;
                    }
                    UAT_DIAG_LAP(kDiagDecode);
                }
            }
        }
        wave_fence();
        // the reference's choice between the two alignments (demod_*_frame at index and index + 1, fewer corrections wins, the first on a tie)
        const int v_take = (skip0 && rs0 <= rs1) ? 0 : (skip1 && rs1 <= rs0) ? 1 : 2;
        const int skip_t = v_take == 0 ? skip0 : v_take == 1 ? skip1 : 0, rs_t = v_take == 0 ? rs0 : rs1;
        uint32_t  up_slot = 0xFFFFFFFFu;
        if (v_take < 2)
        {
            if (kind)
            { // the decoded uplink payload goes to a side array, one 432-byte slot per frame
                uint32_t got = 0;
                if (lane == 0) got = atomicAdd(uplink_count, 1u);
                up_slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                if (up_slot < uplink_cap)
                    for (int k = lane; k < 432; k += 64) uplink_payloads[(size_t)up_slot * 432 + k] = raw[v_take][(k % 72) * 6 + k / 72];
                lane0_join();
            }
            else
            {
                if (lane < 34) pay[lane] = raw[v_take][lane];
                lane0_join();
            }
        }
        const uint64_t after = (kind || skip_t == kUatShortSkip) ? w1 : w2; // what enters the registers after the jump
        if (lane == 0)
        {
            r->index = (uint32_t)idx, r->kind = (uint8_t)kind, r->variant = (uint8_t)v_take;
            r->skip = (int16_t)skip_t, r->rs = (uint8_t)(v_take < 2 ? rs_t : 255);
            r->slot = up_slot;
            if (wn) wn->window = diag::kUatParts <= 6 ? w0 + sink : w0, wn->after = after;
        }
        wave_fence(); // raw[] is reused by the next position
        UAT_DIAG_LAP(kDiagOutput);
        UAT_DIAG_END(kind);
        if (!next_bit) break; // a single look-up: the host follows the loop itself
        if (kind && v_take < 2 && up_slot >= uplink_cap)
        {
            if (lane == 0) counts[kUatCountOverflow] = 1;
            lane0_join();
        }
        if (!chained)
        {
            if (v_take == 2)
            { // no frame at this match: the loop moves on bit by bit
                if (lane == 0) next_bit[c] = 0;
                lane0_join();
                break;
            }
            stale.jump((uint32_t)w0 & kCheckMask, (uint32_t)(w0 >> 32) & kCheckMask, after, (int64_t)sb + skip_t + 1, lenbits, lane);
        }
        else if (v_take < 2) stale.jump(stale.w0_fired, stale.w1_fired, after, (int64_t)sb + skip_t + 1, lenbits, lane);
        // (an extra that does not decode: the loop goes on to the later steps of the same window)
        if (stale.steps == 0)
        { // both registers hold 18 new bits again 17 bits on; a jump past the end of the scanned part stays where it is
            const int64_t nb = stale.bit < lenbits ? (stale.bit + 17 < lenbits ? stale.bit + 17 : lenbits) : stale.bit;
            if (lane == 0) next_bit[c] = (uint32_t)nb;
            lane0_join();
            break;
        }
        // the first step that fires: position and check word as the loop derives them
        const int t = __builtin_ctz(stale.steps);
        stale.steps &= stale.steps - 1;
        stale.w0_fired = (uint32_t)__builtin_amdgcn_readlane((int)stale.w0_step, t), stale.w1_fired = (uint32_t)__builtin_amdgcn_readlane((int)stale.w1_step, t);
        const uint32_t k2  = (stale.w0_fired == kCheckT0 || stale.w1_fired == kCheckT0) ? 0u : 1u;
        const int64_t  sb2 = stale.bit + t - 1 - (kUatCheckBits - 1);
        const uint32_t at  = (uint32_t)(2 * sb2) + (stale.w0_fired == (k2 ? kCheckT1 : kCheckT0) ? 0u : 1u);
        uint32_t       got = 0;
        if (lane == 0) got = atomicAdd(&counts[kUatCountExtras], 1u);
        const uint32_t x = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
        if (x >= extra_cap)
        { // no room: the host will walk the loop itself for this call
            if (lane == 0) counts[kUatCountOverflow] = 1, next_bit[c] = 0;
            lane0_join();
            break;
        }
        if (lane == 0) extras[x].parent = c, extras[x].seq = seq;
        seq++;
        word = at | (k2 << 31), r = &extras[x].rec, wn = nullptr, pay = extra_payloads + (size_t)x * kUatPayloadStride, chained = true;
        }
        item = next_item;
    }
    UAT_DIAG_FLUSH();
}
// ---- ordering: the matches come out of the search in whatever order the waves flushed them; the host walks them in stream
// order.  Counting sort by 32 768-sample bin (matches are sparse: ~2^-17 per sample and check word in noise, a handful per
// bin even in a frame-dense stream), then an insertion sort inside each bin.  Four small launches.
constexpr uint32_t kUatOrderSpanShift = 15;

__global__ __launch_bounds__(256) void uat_order_count_kernel(const uint32_t* __restrict__ cand, uint32_t ncand, uint32_t* __restrict__ span_count)
{
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < ncand; c += gridDim.x * blockDim.x)
        atomicAdd(&span_count[(cand[c] & 0x7FFFFFFFu) >> kUatOrderSpanShift], 1u);
}

// exclusive prefix of span_count[0 .. nspans) in place; one workgroup, every lane owns a contiguous slice
__global__ __launch_bounds__(1024) void uat_order_prefix_kernel(uint32_t* __restrict__ span_count, uint32_t nspans, uint32_t* __restrict__ demod_work,
                                                                uint32_t* __restrict__ up_count)
{
    __shared__ uint32_t partial[1024];
    if (threadIdx.x < kUatDemodRanges) demod_work[threadIdx.x * 32u] = 0; // the demodulation pass that follows starts from zero
    if (threadIdx.x == 0) *up_count = 0;
    const uint32_t      per = (nspans + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < nspans ? lo + per : nspans;
    // a slice that is a whole number of 16-byte groups inside the array goes through registers in one burst of loads (a dependent
    // load per bin made this one-workgroup kernel 20 us long); anything else takes the plain loops
    const bool vec = (per % 4u == 0u) && per <= 32u && lo + per <= nspans;
    uint4      reg[8];
    uint32_t   sum = 0;
    if (vec)
    {
#pragma unroll
        for (uint32_t g = 0; g < 8; g++)
            if (4u * g < per) reg[g] = reinterpret_cast<const uint4*>(span_count + lo)[g];
#pragma unroll
        for (uint32_t g = 0; g < 8; g++)
            if (4u * g < per) sum += reg[g].x + reg[g].y + reg[g].z + reg[g].w;
    }
    else
        for (uint32_t k = lo; k < hi; k++) sum += span_count[k];
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1)
    { // Hillis-Steele inclusive scan
        const uint32_t v = threadIdx.x >= d ? partial[threadIdx.x - d] : 0u;
        __syncthreads();
        partial[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = partial[threadIdx.x] - sum;
    if (vec)
    {
#pragma unroll
        for (uint32_t g = 0; g < 8; g++)
            if (4u * g < per)
            {
                uint4 o;
                o.x = run, run += reg[g].x;
                o.y = run, run += reg[g].y;
                o.z = run, run += reg[g].z;
                o.w = run, run += reg[g].w;
                reinterpret_cast<uint4*>(span_count + lo)[g] = o;
            }
    }
    else
        for (uint32_t k = lo; k < hi; k++)
        {
            const uint32_t c = span_count[k];
            span_count[k]    = run;
            run += c;
        }
}

// span_offset[] is the exclusive prefix; span_fill[] counts what has been placed (zeroed by the caller)
__global__ __launch_bounds__(256) void uat_order_scatter_kernel(const uint32_t* __restrict__ cand, uint32_t ncand, const uint32_t* __restrict__ span_offset,
                                                                uint32_t* __restrict__ span_fill, uint32_t* __restrict__ sorted)
{
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < ncand; c += gridDim.x * blockDim.x)
    {
        const uint32_t v = cand[c], span = (v & 0x7FFFFFFFu) >> kUatOrderSpanShift;
        sorted[span_offset[span] + atomicAdd(&span_fill[span], 1u)] = v;
    }
}

__global__ __launch_bounds__(256) void uat_order_within_kernel(const uint32_t* __restrict__ span_offset, const uint32_t* __restrict__ span_fill,
                                                               uint32_t nspans, uint32_t* __restrict__ sorted, uint32_t* __restrict__ up_list,
                                                               uint32_t* __restrict__ up_count)
{
    // one lane per bin; every wave runs the same number of trips so that the uplink reservation below sees all 64 lanes
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t s0 = blockIdx.x * blockDim.x + threadIdx.x - lane; s0 < nspans; s0 += gridDim.x * blockDim.x)
    {
        const uint32_t s   = s0 + lane;
        const uint32_t n   = s < nspans ? span_fill[s] : 0u;
        const uint32_t off = s < nspans ? span_offset[s] : 0u;
        uint32_t*      a   = sorted + off;
        uint32_t       v[8];
        uint32_t       ups = 0; // uplink matches of this bin
        if (n <= 8)
        { // the usual bin (five matches on a frame-dense stream): all of it in registers with one burst of loads, an odd-even
          // transposition network on the sample index (absent entries sort to the end), one burst of stores -- the in-memory
          // insertion sort below pays a dependent global load per comparison
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) v[i] = i < n ? a[i] : 0xFFFFFFFFu; // no match word is all ones (index < 2^31 - 36)
            auto key = [](uint32_t x) { return x == 0xFFFFFFFFu ? 0xFFFFFFFFu : (x & 0x7FFFFFFFu); };
#pragma unroll
            for (uint32_t round = 0; round < 8; round++)
#pragma unroll
                for (uint32_t i = round & 1u; i + 1 < 8; i += 2)
                {
                    const bool     swap = key(v[i]) > key(v[i + 1]);
                    const uint32_t lo = swap ? v[i + 1] : v[i], hi = swap ? v[i] : v[i + 1];
                    v[i] = lo, v[i + 1] = hi;
                }
#pragma unroll
            for (uint32_t i = 0; i < 8; i++)
                if (i < n)
                {
                    a[i] = v[i];
                    ups += v[i] >> 31;
                }
        }
        else
        {
            for (uint32_t i = 1; i < n; i++)
            { // by sample index; two entries never share one (the check words are complements)
                const uint32_t x = a[i], key = x & 0x7FFFFFFFu;
                uint32_t       j = i;
                for (; j > 0 && (a[j - 1] & 0x7FFFFFFFu) > key; j--) a[j] = a[j - 1];
                a[j] = x;
            }
            for (uint32_t i = 0; i < n; i++) ups += a[i] >> 31;
        }
        // Positions of the uplink matches: the demodulation pass takes these first (they cost ten times an ADS-B match, and one that
        // starts last would be the kernel's tail).  One reservation per wave: an atomic with a reply per match on the one counter
        // (3 700 per GiB) was most of this kernel's 27 us.
        uint32_t incl = ups;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1)
        {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += up;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total == 0) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(up_count, total);
        uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + incl - ups;
        if (n <= 8)
        {
#pragma unroll
            for (uint32_t i = 0; i < 8; i++)
                if (i < n && (v[i] >> 31)) up_list[slot++] = off + i;
        }
        else
            for (uint32_t i = 0; i < n; i++)
                if (a[i] >> 31) up_list[slot++] = off + i;
    }
}

// ---- Round 6: the ordering as ONE launch, from the bins the match search filled (uat_scan_iq_kernel: place_in_bin).  A thread takes one bin:
// the matches of all earlier bins (a sum over the fill counts: the earlier workgroups' by every workgroup for itself, 15 coalesced loads a thread
// at most per GiB; its own by a prefix sum), its <= kUatBinCap matches through a sorting network in registers (Batcher's odd-even merge sort, 63
// compare-exchanges on the match word rotated left by one, so that the order is by sample index), one run of stores.  It also lists the uplink
// matches (which the demodulation takes first), zeroes the demodulation's work counters and the OTHER fill-count array, which the next call fills
// (the two swap roles per call: this call's counts are still being summed by the later workgroups while the earlier ones are done).
struct BatcherPairs
{
    uint8_t a[64], b[64];
    int     n = 0;
    constexpr BatcherPairs()
        : a(), b()
    {
        for (int p = 1; p < (int)kUatBinCap; p <<= 1)
            for (int k = p; k >= 1; k >>= 1)
                for (int j = k % p; j + k < (int)kUatBinCap; j += 2 * k)
                    for (int i = 0; i < k && i + j + k < (int)kUatBinCap; i++)
                        if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) a[n] = (uint8_t)(i + j), b[n] = (uint8_t)(i + j + k), n++;
    }
};
constexpr BatcherPairs kBatcher{};
static_assert(kUatBinCap == 16 && kBatcher.n == 63, "Batcher's network for sixteen");
constexpr int kUatBinThreads = 1024;

__global__ __launch_bounds__(kUatBinThreads) void uat_order_bins_kernel(const uint32_t* __restrict__ bin_fill, uint32_t* __restrict__ bin_fill_next, uint32_t bins_cap,
                                                                        const uint32_t* __restrict__ bin_slots, uint32_t nbins, uint32_t* __restrict__ sorted,
                                                                        uint32_t cap, uint32_t* __restrict__ up_list, uint32_t* __restrict__ up_count,
                                                                        uint32_t* __restrict__ demod_work)
{
    __shared__ uint32_t wave_before[kUatBinThreads / 64], wave_count[kUatBinThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (blockIdx.x == 0 && t < kUatDemodRanges) demod_work[t * 32u] = 0; // the demodulation pass that follows starts from zero
    for (uint32_t k = blockIdx.x * kUatBinThreads + t; k < bins_cap; k += gridDim.x * kUatBinThreads) bin_fill_next[k] = 0;
    const uint32_t bin0 = blockIdx.x * kUatBinThreads, bin = bin0 + t;
    uint32_t       before = 0;
    for (uint32_t b = t; b < bin0; b += kUatBinThreads) before += bin_fill[b]; // (no bin ran over, or the host would not have launched this)
    uint32_t n = bin < nbins ? bin_fill[bin] : 0u;
    n          = n < kUatBinCap ? n : kUatBinCap;
    // the bin's matches, rotated so that an unsigned comparison orders them by sample index; empty slots sort last
    uint32_t v[kUatBinCap];
    {
        const uint4* src = reinterpret_cast<const uint4*>(bin_slots + (size_t)(bin < nbins ? bin : 0u) * kUatBinCap);
#pragma unroll
        for (uint32_t q = 0; q < kUatBinCap / 4; q++)
        {
            const uint4 x = 4 * q < n ? src[q] : make_uint4(0, 0, 0, 0);
            v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
        }
#pragma unroll
        for (uint32_t i = 0; i < kUatBinCap; i++) v[i] = i < n ? ((v[i] << 1) | (v[i] >> 31)) : 0xFFFFFFFFu; // (no match word is all ones)
    }
    // prefix sums: waves first, then the sixteen wave totals
    uint32_t incl = n, binc = before;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1)
    {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += up;
        binc += (uint32_t)__shfl_xor((int)binc, d, 64);
    }
    if (lane == 63) wave_count[wave] = incl, wave_before[wave] = binc;
    __syncthreads();
    uint32_t pos = incl - n;
#pragma unroll
    for (uint32_t w = 0; w < kUatBinThreads / 64; w++) pos += wave_before[w] + (w < wave ? wave_count[w] : 0u);
    if (__ballot(n > 1) != 0)
    {
#pragma unroll
        for (int k = 0; k < kBatcher.n; k++)
        {
            const uint32_t lo = v[kBatcher.a[k]] < v[kBatcher.b[k]] ? v[kBatcher.a[k]] : v[kBatcher.b[k]];
            const uint32_t hi = v[kBatcher.a[k]] < v[kBatcher.b[k]] ? v[kBatcher.b[k]] : v[kBatcher.a[k]];
            v[kBatcher.a[k]] = lo, v[kBatcher.b[k]] = hi;
        }
    }
    uint32_t ups = 0;
#pragma unroll
    for (uint32_t i = 0; i < kUatBinCap; i++)
        if (i < n)
        {
            if (pos + i < cap) sorted[pos + i] = (v[i] >> 1) | (v[i] << 31);
            ups += v[i] & 1u;
        }
    // positions of the uplink matches, one reservation per wave (as uat_order_within_kernel)
    uint32_t uincl = ups;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1)
    {
        const uint32_t up = (uint32_t)__shfl_up((int)uincl, d, 64);
        if (lane >= d) uincl += up;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)uincl, 63);
    if (total == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(up_count, total);
    uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + uincl - ups;
#pragma unroll
    for (uint32_t i = 0; i < kUatBinCap; i++)
        if (i < n && (v[i] & 1u) && pos + i < cap) up_list[slot++] = pos + i;
}

// ---- which frames the scan loop takes.  The loop is sequential -- at the first match it reaches with clean registers it takes the
// frame if one decodes there and jumps (next_bit, which the demodulating wave worked out including any frames behind it), else it
// moves one bit on -- but what it does at a start bit does not depend on how it got there, so the rule is a successor function over
// the ordered matches and the frames taken are the nodes on the path from the first one.  Nodes are the first matches of their
// start bits (a start bit has at most two matches, one per sample alignment: the two check words are complements).  Paths only move
// forward, so the list is cut into blocks of kUatDecideNodes: uat_succ_kernel resolves every node to the first node its path reaches
// outside its block (pointer jumping in LDS, 12 rounds), uat_mark_kernel walks those exits from the first node to its own block (at
// most one dependent load per block) and marks the path inside the block by doubling.
// Threads of a decision workgroup (4096 nodes each).  256 since round 4: with calls in flight the 1024-thread form waited for sixteen free wave
// slots on one CU beside a demodulation kernel that fills every SIMD (13 -> 98 us resident); four waves find room.  Pipelined step 0.540-0.545 ->
// 0.523-0.527 ms, calls back to back 0.750-0.754 -> 0.788-0.794 (the kernels themselves are slower with a quarter of the lanes).
// Round 6: the width is the caller's choice -- 1024 threads for a call that has the chip to itself (13 + 20 us for the two kernels instead of 38 + 35).
constexpr int kUatDecideLevels = 12;
static_assert((1u << kUatDecideLevels) == kUatDecideNodes, "2^levels successors cover a block");
constexpr uint32_t kIndexMask = 0x7FFFFFFFu;

// first position j >= lo with sample index >= key, or n.  The target of an ADS-B jump is a few entries ahead: gallop, then bisect.
__device__ __forceinline__ uint32_t first_at_or_after(const uint32_t* __restrict__ sorted, uint32_t lo, uint32_t n, uint32_t key)
{
    uint32_t hi = lo, step = 1;
    while (hi < n && (sorted[hi] & kIndexMask) < key) lo = hi + 1, hi += step, step <<= 1;
    if (hi > n) hi = n;
    while (lo < hi)
    {
        const uint32_t mid = lo + (hi - lo) / 2;
        if ((sorted[mid] & kIndexMask) < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

template <int kUatDecideThreads>
__global__ __launch_bounds__(kUatDecideThreads) void uat_succ_kernel(const uint32_t* __restrict__ sorted, uint32_t n, const uint32_t* __restrict__ next_bit,
                                                                     int64_t lenbits, int64_t first_bit, int64_t end_bit, uint32_t* __restrict__ succ,
                                                                     uint32_t* __restrict__ exit_of, uint32_t* __restrict__ emit_of, uint32_t* __restrict__ marks)
{
    constexpr int kUatNodesPerLane = (int)kUatDecideNodes / kUatDecideThreads;
    __shared__ uint32_t nxt[kUatDecideNodes];
    const uint32_t      base = blockIdx.x * kUatDecideNodes;
    static_assert(kUatDecideThreads >= (int)kUatDecideNodes / 32 + 1, "a thread per word of the block's bit map");
    if (threadIdx.x < kUatDecideNodes / 32 && base + 32u * threadIdx.x < n) marks[base / 32 + threadIdx.x] = 0; // uat_mark_kernel ORs into it
#pragma unroll
    for (int j = 0; j < kUatNodesPerLane; j++)
    {
        const uint32_t i = threadIdx.x + kUatDecideThreads * j, k = base + i;
        uint32_t       s = kUatEnd, em = kUatEnd;
        if (k < n)
        {
            const uint32_t w = sorted[k], sb = (w & kIndexMask) >> 1;
            const bool     leader = k == 0 || ((sorted[k - 1] & kIndexMask) >> 1) != sb;
            if (leader && (int64_t)sb + (kUatCheckBits - 1) < lenbits && (int64_t)sb < end_bit) // (at or past either the walk has ended here: no successor)
            {
                const bool     two   = k + 1 < n && ((sorted[k + 1] & kIndexMask) >> 1) == sb;
                const uint32_t other = k + 1 + (two ? 1u : 0u); // the next start bit's first match
                // ADS-B before uplink, the even sample before the odd one
                const uint32_t chosen = (two && (w >> 31) && !(sorted[k + 1] >> 31)) ? k + 1 : k;
                const uint32_t nb     = (int64_t)sb >= first_bit ? next_bit[chosen] : 0u; // a start bit before the first one the loop can fire at is passed over
                if (nb == 0) s = other < n ? other : kUatEnd;
                else
                {
                    em = chosen;
                    if ((int64_t)nb < lenbits)
                    { // the loop fires next at the first start bit >= nb - 17
                        const uint32_t j2 = first_at_or_after(sorted, other, n, 2u * (nb - (uint32_t)(kUatCheckBits - 1)));
                        s                 = j2 < n ? j2 : kUatEnd;
                    }
                }
            }
            succ[k] = s, emit_of[k] = em;
        }
        nxt[i] = s;
    }
    __syncthreads();
    for (int level = 0; level < kUatDecideLevels; level++)
    {
        uint32_t v[kUatNodesPerLane];
#pragma unroll
        for (int j = 0; j < kUatNodesPerLane; j++)
        {
            const uint32_t s = nxt[threadIdx.x + kUatDecideThreads * j];
            v[j]             = (s != kUatEnd && s - base < kUatDecideNodes) ? nxt[s - base] : s; // successors lie ahead: s >= base
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kUatNodesPerLane; j++) nxt[threadIdx.x + kUatDecideThreads * j] = v[j];
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < kUatNodesPerLane; j++)
    {
        const uint32_t i = threadIdx.x + kUatDecideThreads * j;
        if (base + i < n) exit_of[base + i] = nxt[i];
    }
}

template <int kUatDecideThreads>
__global__ __launch_bounds__(kUatDecideThreads) void uat_mark_kernel(uint32_t n, const uint32_t* __restrict__ succ, const uint32_t* __restrict__ exit_of,
                                                                     const uint32_t* __restrict__ emit_of, const uint32_t* __restrict__ next_bit,
                                                                     uint32_t* __restrict__ marks, uint32_t* __restrict__ counts)
{
    constexpr int      kUatNodesPerLane = (int)kUatDecideNodes / kUatDecideThreads;
    constexpr uint16_t kOut = 0xFFFFu;
    __shared__ uint16_t hop[2][kUatDecideNodes]; // hop[l & 1][i]: the node 2^l steps after i, kOut = outside the block
    __shared__ uint8_t  on_path[kUatDecideNodes];
    const uint32_t      base = blockIdx.x * kUatDecideNodes;
    // where the path from the first match enters this block (the same walk in every lane: a chain of at most blockIdx.x loads)
    uint32_t e = 0;
    while (e != kUatEnd && e < base) e = exit_of[e];
    if (e == kUatEnd || e - base >= kUatDecideNodes) return; // the path passes this block by (marks[] was zeroed by uat_succ_kernel)
#pragma unroll
    for (int j = 0; j < kUatNodesPerLane; j++)
    {
        const uint32_t i = threadIdx.x + kUatDecideThreads * j;
        const uint32_t s = base + i < n ? succ[base + i] : kUatEnd;
        hop[0][i]        = (s != kUatEnd && s - base < kUatDecideNodes) ? (uint16_t)(s - base) : kOut;
        on_path[i]       = base + i == e;
    }
    __syncthreads();
    // Doubling, marks and steps together: before level l the marks cover the distances 0 .. 2^l - 1 from the entry and hop[l & 1] is the
    // 2^l-step table; marking what lies 2^l steps behind every marked node doubles the range, squaring the table gives the next one.
    // (A mark another lane sets in the same level only adds nodes that are on the path as well.)  20 KB of LDS instead of the 96 KB
    // of twelve stored tables: the workgroup fits beside a scan kernel's on the same CU.
    for (int level = 0; level < kUatDecideLevels; level++)
    {
        const uint16_t* cur = hop[level & 1];
        uint16_t*       nxt = hop[(level & 1) ^ 1];
#pragma unroll
        for (int j = 0; j < kUatNodesPerLane; j++)
        {
            const uint32_t i = threadIdx.x + kUatDecideThreads * j;
            const uint16_t h = cur[i];
            if (h != kOut && on_path[i]) on_path[h] = 1;
            nxt[i] = h == kOut ? kOut : cur[h];
        }
        __syncthreads();
    }
    // The bit map of this block's matches is put together in LDS and goes out with one atomic per word: device-scope atomics are
    // performed past the XCD's L2 (one per frame taken, 74 000 per GiB, made this kernel 64 us long).  A start bit's second match
    // can be the first match of the next block, hence one word more than the block has, and atomics rather than stores.
    // (It lives where the hop tables were -- nobody reads them after the last level: 20 480 bytes in all, which fits beside a scan kernel's
    // workgroup on a CU.  Round 4's demodulation kernel left that much free too; round 5's takes the whole LDS while it runs.)
    uint32_t* const map = reinterpret_cast<uint32_t*>(&hop[0][0]);
    static_assert(sizeof(hop) >= (kUatDecideNodes / 32 + 1) * sizeof(uint32_t), "the bit map fits where the tables were");
    if (threadIdx.x < kUatDecideNodes / 32 + 1) map[threadIdx.x] = 0;
    __syncthreads();
    uint32_t taken = 0, last = 0;
#pragma unroll
    for (int j = 0; j < kUatNodesPerLane; j++)
    {
        const uint32_t i = threadIdx.x + kUatDecideThreads * j;
        if (!on_path[i]) continue;
        const uint32_t em = emit_of[base + i];
        if (em == kUatEnd) continue;
        atomicOr(&map[(em - base) >> 5], 1u << (em & 31u)); // base is a multiple of 32
        const uint32_t nb = next_bit[em];
        last              = nb > last ? nb : last;
        taken++;
    }
    __syncthreads();
    if (threadIdx.x < kUatDecideNodes / 32 + 1 && map[threadIdx.x]) atomicOr(&marks[base / 32 + threadIdx.x], map[threadIdx.x]);
    // one pair of atomics per wave
    for (int d = 32; d >= 1; d >>= 1)
    {
        taken += (uint32_t)__shfl_xor((int)taken, d, 64);
        const uint32_t o = (uint32_t)__shfl_xor((int)last, d, 64);
        last             = o > last ? o : last;
    }
    if ((threadIdx.x & 63) == 0 && taken)
    {
        atomicAdd(&counts[kUatCountTaken], taken);
        atomicMax(&counts[kUatCountFinalBit], last);
    }
}

} // namespace

hipError_t launch_uat978_decide(const UatArgs& a, uint32_t ncand, const uint32_t* sorted, hipStream_t stream, bool wide)
{
    if (ncand == 0 || a.lenbits <= 0) return hipSuccess;
    const uint32_t blocks = (ncand + kUatDecideNodes - 1) / kUatDecideNodes;
    if (wide)
    {
        hipLaunchKernelGGL(uat_succ_kernel<1024>, dim3(blocks), dim3(1024), 0, stream, sorted, ncand, a.next_bit, a.lenbits, a.first_bit, a.end_bit, a.succ, a.exit_of, a.emit_of, a.marks);
        hipLaunchKernelGGL(uat_mark_kernel<1024>, dim3(blocks), dim3(1024), 0, stream, ncand, a.succ, a.exit_of, a.emit_of, a.next_bit, a.marks, a.counts);
    }
    else
    {
        hipLaunchKernelGGL(uat_succ_kernel<256>, dim3(blocks), dim3(256), 0, stream, sorted, ncand, a.next_bit, a.lenbits, a.first_bit, a.end_bit, a.succ, a.exit_of, a.emit_of, a.marks);
        hipLaunchKernelGGL(uat_mark_kernel<256>, dim3(blocks), dim3(256), 0, stream, ncand, a.succ, a.exit_of, a.emit_of, a.next_bit, a.marks, a.counts);
    }
    return hipGetLastError();
}

hipError_t launch_uat978(const UatArgs& a, hipStream_t stream)
{
    if (a.nsamples < 2) return hipMemsetAsync(a.counts, 0, kUatCountWords * sizeof(uint32_t), stream);
    hipError_t e = hipMemsetAsync(a.counts, 0, kUatCountWords * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    if (!a.phases_given)
    {
        const uint64_t nwgs = ((a.nsamples + kUatWaveSamples - 1) / kUatWaveSamples + kUatScanWaves - 1) / kUatScanWaves;
        const uint32_t grid = (uint32_t)(nwgs < 256 ? nwgs : 256); // one workgroup per CU (its LDS footprint allows no more)
        hipLaunchKernelGGL(uat_scan_iq_kernel, dim3(grid), dim3(kUatScanThreads), 0, stream, a.in, a.lut, a.nsamples, a.cand, a.cand_cap, a.counts, a.bin_fill, a.bin_slots);
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

// ordered: the list is the output of launch_uat978_order on the same stream (which also zeroed the work counters and collected
// the uplink positions, a.up_list / a.counts + 2); otherwise a handful of look-ups the host asked for
hipError_t launch_uat978_demod(const UatArgs& a, uint32_t ncand, bool ordered, hipStream_t stream)
{
    if (ncand == 0) return hipSuccess;
    uint32_t* const work = a.demod_work;
    const uint32_t nranges = ncand >= 4096 ? kUatDemodRanges : 1u;
    // one wave per match up to what the device holds at once (two workgroups per CU: the waves draw tickets from there on)
    const uint32_t kResidentWaves = 2 * (a.ncu ? a.ncu : 256u) * kUatDemodWaves;
    uint32_t waves = ncand > kResidentWaves ? kResidentWaves : ncand;
    waves          = ((waves + nranges - 1) / nranges) * nranges;
    const uint32_t g = (waves + kUatDemodWaves - 1) / kUatDemodWaves;
    if (!ordered && ncand > 1)
    { // (a single look-up needs no reset: one wave, one item, and the loop ends whatever the counter holds; the next ordering pass zeroes it)
        hipError_t e = hipMemsetAsync(work, 0, kUatDemodRanges * 32 * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    const uint32_t* up_list  = ordered ? a.up_list : nullptr;
    const uint32_t* up_count = ordered ? a.counts + 2 : nullptr;
    uint32_t*       chase    = ordered ? a.next_bit : nullptr; // the frames behind a frame are followed for the ordered list only
    if (a.phases_given)
        hipLaunchKernelGGL(uat_demod_kernel<true>, dim3(g), dim3(64 * kUatDemodWaves), 0, stream, a.in, a.lut, a.nsamples, a.rs_tables, a.cand, ncand, a.recs, a.wins,
                           a.payloads, a.uplink_payloads, a.uplink_cap, a.counts + 1, work, nranges, up_list, up_count, a.single_word,
                           a.lenbits, chase, a.extras, a.extra_payloads, a.extra_cap, a.counts);
    else
        hipLaunchKernelGGL(uat_demod_kernel<false>, dim3(g), dim3(64 * kUatDemodWaves), 0, stream, a.in, a.lut, a.nsamples, a.rs_tables, a.cand, ncand, a.recs, a.wins,
                           a.payloads, a.uplink_payloads, a.uplink_cap, a.counts + 1, work, nranges, up_list, up_count, a.single_word,
                           a.lenbits, chase, a.extras, a.extra_payloads, a.extra_cap, a.counts);
    return hipGetLastError();
}

// a.cand[0 .. ncand) -> sorted[0 .. ncand) by sample index; positions of the uplink matches in `sorted` -> a.up_list, their number ->
// a.counts[2]; the demodulation work counters zeroed.  scratch: 2 * nspans words, nspans = spans covering a.nsamples.
hipError_t launch_uat978_order(const UatArgs& a, uint32_t ncand, uint32_t* scratch, uint32_t* sorted, hipStream_t stream)
{
    if (ncand == 0) return hipSuccess;
    const uint32_t nspans = (uint32_t)((a.nsamples + (1u << kUatOrderSpanShift) - 1) >> kUatOrderSpanShift);
    uint32_t*      offset = scratch;
    uint32_t*      fill   = scratch + nspans;
    hipError_t     e      = hipMemsetAsync(scratch, 0, 2 * (size_t)nspans * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const uint32_t gc = (ncand + 255) / 256 > 1024 ? 1024 : (ncand + 255) / 256;
    const uint32_t gs = (nspans + 255) / 256 > 1024 ? 1024 : (nspans + 255) / 256;
    hipLaunchKernelGGL(uat_order_count_kernel, dim3(gc), dim3(256), 0, stream, a.cand, ncand, offset);
    hipLaunchKernelGGL(uat_order_prefix_kernel, dim3(1), dim3(1024), 0, stream, offset, nspans, a.demod_work, a.counts + 2);
    hipLaunchKernelGGL(uat_order_scatter_kernel, dim3(gc), dim3(256), 0, stream, a.cand, ncand, offset, fill, sorted);
    hipLaunchKernelGGL(uat_order_within_kernel, dim3(gs), dim3(256), 0, stream, offset, fill, nspans, sorted, a.up_list, a.counts + 2);
    return hipGetLastError();
}

hipError_t launch_uat978_order_bins(const UatArgs& a, uint32_t ncand, uint32_t* sorted, hipStream_t stream)
{
    if (ncand == 0) return hipSuccess; // (nothing was placed: both fill-count arrays are still all zero)
    const uint32_t nbins = (uint32_t)((a.nsamples + (1u << kUatBinShift) - 1) >> kUatBinShift);
    hipLaunchKernelGGL(uat_order_bins_kernel, dim3((nbins + kUatBinThreads - 1) / kUatBinThreads), dim3(kUatBinThreads), 0, stream, a.bin_fill, a.bin_fill_next,
                       a.bins_cap, a.bin_slots, nbins, sorted, a.cand_cap, a.up_list, a.counts + 2, a.demod_work);
    return hipGetLastError();
}

#if DIAG_UAT
extern "C" int adsb_amd_uat_diag(unsigned long long* out16)
{ // sums since the last call; the launches must have completed
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_uat_diag), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_uat_diag), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_uat978_rs_selftest(const RsTables* tables, int kind, uint8_t* words, int* results, int count, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(uat_rs_selftest_kernel, dim3(count > 4096 ? 4096 : count), dim3(64), 0, stream, tables, kind, words, results, count);
    return hipGetLastError();
}

} // namespace adsb_amd
