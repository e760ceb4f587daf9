// scan1090.hip -- gfx950 (MI355X, CDNA4) kernels for the 1090ES IQ -> Mode S candidate-record path.
//
// One wavefront (a 64-thread workgroup) owns one "chunk": 4096 consecutive preamble positions of one
// reference buffer.  It streams the 8.5 KiB of u8 IQ that those positions can touch with 16-byte
// coalesced loads, turns every I,Q pair into s = (I-127)^2 + (Q-127)^2 with packed 16-bit math and
// parks s (u16) in a wave-private LDS window.  Everything after that works out of LDS:
//
//   stage 1  (reference ADSB1090.cpp:782-783)  ten preamble comparisons per position, evaluated on s
//            instead of on the magnitude: the reference LUT m = round(360*sqrt(s)) is strictly
//            increasing on every reachable s, so <,> between magnitudes equal <,> between s values.
//            Packed: two positions per VALU op, conditions folded into four "low < high" bounds.
//   stage 2  (:794-811) exact magnitudes (LUT by s) for the ~1 % of positions that survive, dense over lanes.
//   demod    (:814-881, 277-332) one candidate at a time, all 64 lanes: lane b slices bit b (and 64+b),
//            the "copy the previous bit" and the phase-correction recurrences are resolved with
//            __ballot masks, parity is a wave XOR of per-lane table entries, the 1-bit repair is a
//            ballot over per-lane syndromes.
//
// The sequential part of the reference (skip-ahead after an accepted frame, ICAO-cache gating of
// AP-type DFs; :886-957) is NOT done here: the kernel emits one record per (offset, pass) that the
// reference could accept and the host resolver applies those rules in sample order.
//
// HBM traffic: the IQ bytes once (+6 % halo, normally an L2 hit: neighbouring chunks are scheduled on
// the same XCD) and 32 B per record.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>

#include "scan1090.h"

namespace adsb_amd
{
namespace
{

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2    as_pk(uint32_t x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ uint32_t as_u32(u16x2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_max(as_pk(a), as_pk(b))); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_min(as_pk(a), as_pk(b))); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u32(as_pk(a) - as_pk(b)); }

// Two IQ samples (4 bytes I0 Q0 I1 Q1) -> (s0 | s1 << 16), s = (I-127)^2 + (Q-127)^2 clamped to 32767.
// 32768 (I=Q=255... i=q=128) is the only value above 32767 and 32767 itself is not a sum of two squares,
// so the clamp keeps the order of all reachable values and makes every difference fit in an int16.
__device__ __forceinline__ uint32_t iq2_to_s2(uint32_t x)
{
    u16x2 i  = as_pk(x & 0x00FF00FFu);
    u16x2 q  = as_pk((x >> 8) & 0x00FF00FFu);
    u16x2 c  = {127, 127};
    u16x2 di = i - c;
    u16x2 dq = q - c;
    u16x2 s  = di * di + dq * dq;
    u16x2 lim = {32767, 32767};
    return as_u32(__builtin_elementwise_min(s, lim));
}

__device__ __forceinline__ uint32_t iq1_to_s(uint32_t i, uint32_t q)
{
    int      di = (int)i - 127, dq = (int)q - 127;
    uint32_t s  = (uint32_t)(di * di + dq * dq);
    return s > 32767u ? 32767u : s;
}

__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint32_t wave_xor(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v ^= (uint32_t)__shfl_xor((int)v, o);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// exclusive prefix sum over the 64 lanes, *total receives the wave sum
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane, uint32_t* total)
{
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
        uint32_t y = (uint32_t)__shfl_up((int)x, o);
        if (lane >= o) x += y;
    }
    *total = (uint32_t)__shfl((int)x, 63);
    return x - v;
}

// mask of bits 0..b (b in 0..63)
__device__ __forceinline__ uint64_t upto(int b) { return (b >= 63) ? ~0ull : ((2ull << b) - 1ull); }

// Value of a run-length "hold" sequence at bit b: the value at the highest decided position <= b,
// or `carry` when no position <= b is decided.
__device__ __forceinline__ uint32_t hold_resolve(uint64_t decided, uint64_t value, int b, uint32_t carry)
{
    uint64_t m = decided & upto(b);
    if (m == 0) return carry;
    int p = 63 - __builtin_clzll(m);
    return (uint32_t)((value >> p) & 1ull);
}

// Two-state chain t_k = t_{k-1} ? up_k : dn_k resolved from ballots: `cst` marks positions where up==dn
// (value in `val`), `inv` marks positions where the state is inverted (up=0, dn=1); elsewhere identity.
__device__ __forceinline__ uint32_t chain_resolve(uint64_t cst, uint64_t val, uint64_t inv, int b, uint32_t carry)
{
    uint64_t m = cst & upto(b);
    uint32_t start;
    uint64_t span;
    if (m == 0)
    {
        start = carry;
        span  = upto(b);
    }
    else
    {
        int p = 63 - __builtin_clzll(m);
        start = (uint32_t)((val >> p) & 1ull);
        span  = upto(b) & ~upto(p);
    }
    return start ^ (uint32_t)(__builtin_popcountll(inv & span) & 1);
}

struct LaneTables
{
    uint32_t syn_a;  // 112-bit message: syndrome of flipping bit `lane`
    uint32_t syn_b;  // 112-bit message: syndrome of flipping bit 64+lane (lane < 48)
    uint32_t syn_s;  // 56-bit message: syndrome of flipping bit `lane` (lane < 56)
    uint32_t crc_a;  // parity-table entry of bit `lane` (112-bit)
    uint32_t crc_b;  // parity-table entry of bit 64+lane
    uint32_t crc_s;  // parity-table entry of bit `lane` of a 56-bit message
};

__device__ __forceinline__ LaneTables load_lane_tables(const uint32_t* __restrict__ tab, int lane)
{
    LaneTables t;
    // ModesChecksumTable semantics (ADSB1090.cpp:266-275): entry b < 88 = x^(111-b) mod G, last 24 entries 0.
    uint32_t ta = tab[lane];
    uint32_t tb = (lane < 48) ? tab[64 + lane] : 0u;
    uint32_t ts = (lane < 56) ? tab[56 + lane] : 0u;
    t.crc_a = ta;
    t.crc_b = tb;
    t.crc_s = ts;
    // flipping a bit of the parity field itself changes the stored value by that bit (FixSingleBitErrors :304-332)
    int ba = lane, bb = 64 + lane;
    t.syn_a = ta; // lane < 64 < 88: always a data bit
    t.syn_b = (lane < 48) ? ((bb < 88) ? tb : (1u << (111 - bb))) : 0xFFFFFFFFu;
    t.syn_s = (lane < 56) ? ((ba < 32) ? ts : (1u << (55 - ba))) : 0xFFFFFFFFu;
    return t;
}

struct Emit
{
    adsb_amd_record_t* base; // this chunk's region
    uint32_t           cap;
    uint32_t           count; // wave-uniform
    uint32_t           buffer;
};

__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// Store one record (wave-uniform arguments, lane 0 writes 32 bytes).
__device__ __forceinline__ void emit_record(Emit& e, int lane, uint32_t offset, uint64_t ra, uint64_t rb, uint32_t df, uint32_t nbits,
                                            int errorbit, uint32_t flags, uint32_t addr, uint32_t delta)
{
    if (e.count < e.cap)
    {
        if (lane == 0)
        {
            uint32_t m0 = bswap32((uint32_t)(ra >> 32)), m1 = bswap32((uint32_t)ra);
            uint32_t m2 = bswap32((uint32_t)(rb >> 32)), m3 = bswap32((uint32_t)rb);
            uint32_t d16 = delta > 65535u ? 65535u : delta;
            uint4    lo, hi;
            lo.x = e.buffer;
            lo.y = offset;
            lo.z = addr;
            lo.w = d16 | (nbits << 16) | (((uint32_t)errorbit & 0xFFu) << 24);
            hi.x = df | (flags << 8) | (m0 << 16);
            hi.y = (m0 >> 16) | (m1 << 16);
            hi.z = (m1 >> 16) | (m2 << 16);
            hi.w = (m2 >> 16) | (m3 << 16);
            uint4* dst = reinterpret_cast<uint4*>(e.base + e.count);
            dst[0]     = lo;
            dst[1]     = hi;
        }
    }
    e.count++; // counts past cap signal overflow to the prefix kernel
}

__device__ __forceinline__ bool df_is_long(uint32_t df) { return df == 16 || df == 17 || df == 19 || df == 20 || df == 21; }
__device__ __forceinline__ bool df_is_ap(uint32_t df) { return df == 0 || df == 4 || df == 5 || df == 16 || df == 20 || df == 21 || df == 24; }

// Parity / repair / classification of one sliced message (wave-uniform ba/bb = message bits 0..63 / 64..111,
// LSB = lowest bit index).  Returns 0: nothing to emit, 1: emitted.  *stateless is set when the frame is one the
// reference accepts without consulting the ICAO cache (DF11/17 with good or repaired parity).
__device__ __forceinline__ int classify_and_emit(Emit& e, int lane, const LaneTables& lt, uint64_t ba, uint64_t bb, uint32_t df,
                                                 uint32_t nbits, uint32_t offset, uint32_t flags, uint32_t delta, bool* stateless)
{
    *stateless      = false;
    const bool is17 = (df == 11 || df == 17);
    if (!is17 && !df_is_ap(df)) return 0; // BruteForceAp returns 0 for every other DF (:403-409)

    const bool bit_a = (ba >> lane) & 1ull;
    const bool bit_b = (lane < 48) && ((bb >> lane) & 1ull);
    uint32_t   contrib, stored;
    uint64_t   ra = __builtin_bitreverse64(ba), rb = __builtin_bitreverse64(bb);
    if (nbits == 112)
    {
        contrib = (bit_a ? lt.crc_a : 0u) ^ (bit_b ? lt.crc_b : 0u);
        stored  = (uint32_t)(rb >> 16) & 0xFFFFFFu;
    }
    else
    {
        contrib = (bit_a && lane < 56) ? lt.crc_s : 0u;
        stored  = (uint32_t)(ra >> 8) & 0xFFFFFFu;
    }
    const uint32_t crc = wave_xor(contrib);
    const uint32_t syn = crc ^ stored;
    if (is17)
    {
        int errorbit = -1;
        if (syn != 0)
        {
            // first bit (ascending) whose flip makes stored == computed
            if (nbits == 112)
            {
                uint64_t ma = ballot(syn == lt.syn_a);
                uint64_t mb = ballot(lane < 48 && syn == lt.syn_b);
                if (ma) errorbit = __builtin_ctzll(ma);
                else if (mb) errorbit = 64 + __builtin_ctzll(mb);
            }
            else
            {
                uint64_t ms = ballot(lane < 56 && syn == lt.syn_s);
                if (ms) errorbit = __builtin_ctzll(ms);
            }
            if (errorbit < 0) return 0;
            if (errorbit < 64) ba ^= (1ull << errorbit);
            else bb ^= (1ull << (errorbit - 64));
            ra = __builtin_bitreverse64(ba);
            rb = __builtin_bitreverse64(bb);
        }
        *stateless    = true;
        uint32_t addr = (uint32_t)(ra >> 32) & 0xFFFFFFu;
        emit_record(e, lane, offset, ra, rb, df, nbits, errorbit, flags, addr, delta);
        return 1;
    }
    // AP-type: address candidate = AP xor parity (:418-425); validity is decided against the ICAO cache on the host
    emit_record(e, lane, offset, ra, rb, df, nbits, -1, flags | ADSB_AMD_F_NEEDS_ICAO, syn, delta);
    return 1;
}

// Demodulate the candidate whose preamble starts at tile index w0 (sample j of the buffer).
__device__ __forceinline__ void demod_candidate(const uint16_t* tile, const uint16_t* __restrict__ lut, int lane, const LaneTables& lt,
                                                Emit& e, int w0, uint32_t j)
{
    const bool has_b = lane < 48;
    // bit `lane` lives in samples j+16+2*lane, j+17+2*lane; bit 64+lane another 128 samples on (ADSB1090.cpp:831-835)
    const int ia  = w0 + 16 + 2 * lane;
    const int ib  = has_b ? ia + 128 : ia;
    const int loA = lut[tile[ia]], hiA = lut[tile[ia + 1]];
    const int loB = lut[tile[ib]], hiB = lut[tile[ib + 1]];
    // preamble neighbourhood m[j-1 .. j+14] on lanes 0..15 (DetectOutOfPhase :683-690)
    const int pre = lut[tile[w0 - 1 + (lane & 15)]];

    const int dA = abs(loA - hiA);
    const int dB = has_b ? abs(loB - hiB) : 0;

    // bit 0 with equal halves is the reference's only reachable "errors++" (:839-846); it survives the retry
    // unchanged (sample j+16 is never rescaled), so such a candidate can never be decoded.
    if (__builtin_amdgcn_readfirstlane((int)(loA == hiA))) return;

    // energy sums over the untouched samples (:870-872): first 56 bits, and the rest
    const uint32_t sum56   = wave_sum(lane < 56 ? (uint32_t)dA : 0u);
    const uint32_t sumrest = wave_sum((lane >= 56 ? (uint32_t)dA : 0u) + (uint32_t)dB);

    // ---------------- pass 1: plain slice (:831-853)
    const uint64_t decA  = ballot(lane == 0 || dA >= 256);
    const uint64_t valA  = ballot(loA > hiA);
    const uint64_t decB  = ballot(has_b && dB >= 256);
    const uint64_t valB  = ballot(has_b && loB > hiB);
    const uint32_t bitA  = hold_resolve(decA, valA, lane, 0u);
    const uint32_t last  = hold_resolve(decA, valA, 63, 0u);
    const uint32_t bitB  = hold_resolve(decB, valB, lane, last);
    uint64_t       ba    = ballot(bitA != 0);
    uint64_t       bb    = ballot(has_b && bitB != 0);
    uint32_t       df    = (uint32_t)(__builtin_bitreverse64(ba) >> 59);
    uint32_t       nbits = df_is_long(df) ? 112u : 56u;
    uint32_t       delta = (nbits == 112u) ? (sum56 + sumrest) / 56u : sum56 / 28u;
    if (delta < 2550u) return; // :877-881, no retry either
    bool stateless = false;
    classify_and_emit(e, lane, lt, ba, bb, df, nbits, j, 0u, delta, &stateless);
    if (stateless) return; // the reference accepts here and never retries

    // ---------------- pass 2: retry with phase correction (:814-826); identical to pass 1 unless the window is rescaled
    if (j == 0) return;
    const int m_1 = __shfl(pre, 0), m1 = __shfl(pre, 2), m2 = __shfl(pre, 3), m3 = __shfl(pre, 4);
    const int m6 = __shfl(pre, 7), m7 = __shfl(pre, 8), m9 = __shfl(pre, 10), m10 = __shfl(pre, 11);
    // x > y/3 (integer division)  <=>  3x > y
    const bool oop = (3 * m3 > m2) || (3 * m10 > m9) || (3 * m6 > m7) || (3 * m_1 > m1);
    if (!oop) return;

    // ApplyPhaseCorrection (:720-736): the first sample of bit k>=1 becomes up(x) or dn(x) (u16 wrap) depending on
    // whether the (already rescaled) first sample of bit k-1 exceeds its second sample.
    const int  upA = (int)(uint16_t)((loA * 5) / 4), dnA = (int)(uint16_t)((loA * 4) / 5);
    const int  upB = (int)(uint16_t)((loB * 5) / 4), dnB = (int)(uint16_t)((loB * 4) / 5);
    const bool cuA = (lane == 0) ? (loA > hiA) : (upA > hiA);
    const bool cdA = (lane == 0) ? (loA > hiA) : (dnA > hiA);
    const bool cuB = has_b && (upB > hiB);
    const bool cdB = has_b && (dnB > hiB);
    const uint64_t cstA = ballot(cuA == cdA), vlA = ballot(cuA), ivA = ballot(!cuA && cdA);
    const uint64_t cstB = ballot(has_b && cuB == cdB), vlB = ballot(cuB), ivB = ballot(has_b && !cuB && cdB);
    const uint32_t tA   = chain_resolve(cstA, vlA, ivA, lane, 0u);
    const uint32_t t63  = chain_resolve(cstA, vlA, ivA, 63, 0u);
    const uint32_t tB   = chain_resolve(cstB, vlB, ivB, lane, t63);
    const uint64_t TA   = ballot(tA != 0);
    const uint64_t TB   = ballot(has_b && tB != 0);
    const uint32_t prevA = (lane == 0) ? 0u : (uint32_t)((TA >> (lane - 1)) & 1ull);
    const uint32_t prevB = (lane == 0) ? t63 : (uint32_t)((TB >> (lane - 1)) & 1ull);
    const int      lo2A  = (lane == 0) ? loA : (prevA ? upA : dnA);
    const int      lo2B  = prevB ? upB : dnB;
    const int      d2A   = abs(lo2A - hiA);
    const int      d2B   = has_b ? abs(lo2B - hiB) : 0;
    const uint64_t dec2A = ballot(lane == 0 || d2A >= 256);
    const uint64_t val2A = ballot(lo2A > hiA);
    const uint64_t dec2B = ballot(has_b && d2B >= 256);
    const uint64_t val2B = ballot(has_b && lo2B > hiB);
    const uint32_t bit2A = hold_resolve(dec2A, val2A, lane, 0u);
    const uint32_t last2 = hold_resolve(dec2A, val2A, 63, 0u);
    const uint32_t bit2B = hold_resolve(dec2B, val2B, lane, last2);
    ba    = ballot(bit2A != 0);
    bb    = ballot(has_b && bit2B != 0);
    df    = (uint32_t)(__builtin_bitreverse64(ba) >> 59);
    nbits = df_is_long(df) ? 112u : 56u;
    delta = (nbits == 112u) ? (sum56 + sumrest) / 56u : sum56 / 28u; // window is restored before the gate (:855-856)
    if (delta < 2550u) return;
    classify_and_emit(e, lane, lt, ba, bb, df, nbits, j, ADSB_AMD_F_PASS2 | ADSB_AMD_F_PHASE, delta, &stateless);
}

// 16 bytes of IQ at sample g of the buffer, zero beyond the buffer end.
__device__ __forceinline__ uint4 load_iq16(const uint8_t* __restrict__ buf, uint32_t g, uint32_t n)
{
    if (g + 8u <= n) return *reinterpret_cast<const uint4*>(buf + 2ull * g);
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (uint32_t k = 0; k < 8u; k++)
    {
        if (g + k < n)
        {
            uint32_t v = (uint32_t)buf[2ull * (g + k)] | ((uint32_t)buf[2ull * (g + k) + 1] << 8);
            w[k >> 1] |= v << (16u * (k & 1u));
        }
        else
        {
            // sample beyond the end: s = 0 (I = Q = 127); never read by a valid position
            w[k >> 1] |= 0x7F7Fu << (16u * (k & 1u));
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(64) void scan1090_kernel(ScanArgs a)
{
    __shared__ __attribute__((aligned(16))) uint16_t tile[kTileU16];
    __shared__ uint16_t                              queue[kChunk / 2]; // stage-1 survivors (adjacent positions cannot both pass)

    const int        lane = threadIdx.x;
    const LaneTables lt   = load_lane_tables(a.crc_tab, lane);

    // XCD-aware chunk order: workgroups b and b+8 share an XCD (round-robin dispatch), so give every XCD one
    // contiguous range of chunks and let its workgroups walk that range together -> halo re-reads hit its L2.
    const uint32_t nxcd  = 8;
    const uint32_t xcd   = blockIdx.x % nxcd;
    const uint32_t slot  = blockIdx.x / nxcd;
    const uint32_t nslot = gridDim.x / nxcd; // grid is a multiple of 8
    const uint32_t per   = (a.total_chunks + nxcd - 1) / nxcd;
    const uint32_t first = xcd * per;
    const uint32_t end   = (first + per < a.total_chunks) ? first + per : a.total_chunks;

    for (uint32_t chunk = first + slot; chunk < end; chunk += nslot)
    {
        const uint32_t bidx  = chunk / a.chunks_per_buf;
        const uint32_t cidx  = chunk - bidx * a.chunks_per_buf;
        const uint32_t n     = a.buf_samples;
        const uint32_t g0    = cidx * (uint32_t)kChunk;
        const uint32_t limit = n - (uint32_t)kFrameSpan; // positions j < limit
        const uint32_t npos  = (limit - g0 < (uint32_t)kChunk) ? (limit - g0) : (uint32_t)kChunk;
        const uint8_t* buf   = a.iq + (uint64_t)bidx * a.buf_stride;

        // ---------------- load the window (8.5 rows of 1 KiB) and park s in LDS
        uint4 raw[kRows + 1];
#pragma unroll
        for (int r = 0; r <= kRows; r++)
        {
            const uint32_t g = g0 + (uint32_t)(r * kRowSamples + 8 * lane);
            if (r < kRows || lane < 32) raw[r] = load_iq16(buf, g, n);
            else raw[r] = make_uint4(0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu);
        }
        uint32_t sfront = 0;
        if (lane == 0 && g0 > 0) sfront = iq1_to_s(buf[2ull * (g0 - 1)], buf[2ull * (g0 - 1) + 1]);

        __syncthreads(); // previous chunk's readers are done with the tile (single-wave workgroup: no s_barrier cost)
        uint4 sown[kRows];
#pragma unroll
        for (int r = 0; r <= kRows; r++)
        {
            uint4 s;
            s.x = iq2_to_s2(raw[r].x);
            s.y = iq2_to_s2(raw[r].y);
            s.z = iq2_to_s2(raw[r].z);
            s.w = iq2_to_s2(raw[r].w);
            if (r < kRows) sown[r] = s;
            if (r < kRows || lane < 32) *reinterpret_cast<uint4*>(&tile[kFront + r * kRowSamples + 8 * lane]) = s;
        }
        if (lane == 0) tile[kFront - 1] = (uint16_t)sfront;
        __syncthreads();

        // ---------------- stage 1 on packed s: positions (2i, 2i+1) of this lane's 8, i = 0..3
        uint64_t surv = 0;
#pragma unroll
        for (int r = 0; r < kRows; r++)
        {
            uint32_t    p0[9];
            const int   w  = kFront + r * kRowSamples + 8 * lane;
            const uint4 nx = *reinterpret_cast<const uint4*>(&tile[w + 8]);
            p0[0] = sown[r].x; p0[1] = sown[r].y; p0[2] = sown[r].z; p0[3] = sown[r].w;
            p0[4] = nx.x; p0[5] = nx.y; p0[6] = nx.z; p0[7] = nx.w;
            p0[8] = *reinterpret_cast<const uint32_t*>(&tile[w + 16]);
            uint32_t p1[8]; // odd-aligned pairs (s[2i+1], s[2i+2])
#pragma unroll
            for (int i = 0; i < 8; i++) p1[i] = __builtin_amdgcn_alignbit(p0[i + 1], p0[i], 16);
            uint32_t m2o[6]; // (max(s[2i+1],s[2i+2]), max(s[2i+2],s[2i+3]))
#pragma unroll
            for (int i = 1; i <= 5; i++) m2o[i] = pk_max(p1[i], p0[i + 1]);
            uint32_t bits = 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                // with j = 2i (low half) / 2i+1 (high half), m_k = s[j+k]:
                const uint32_t mx36 = pk_max(m2o[i + 1], m2o[i + 2]);             // max(m3..m6)
                const uint32_t l0   = pk_max(p1[i], mx36);                        // max(m1, m3..m6)
                const uint32_t d1   = pk_sub(l0, p0[i]);                          // < 0 : all of them < m0
                const uint32_t d2   = pk_sub(pk_max(p1[i], p1[i + 1]), p0[i + 1]); // max(m1,m3) < m2
                const uint32_t d3   = pk_sub(p0[i + 4], pk_min(p1[i + 3], p1[i + 4])); // m8 < min(m7,m9)
                const uint32_t d4   = pk_sub(p0[i + 3], p1[i + 4]);               // m6 < m9
                const uint32_t ok   = (d1 & d2 & d3 & d4) & 0x80008000u;
                bits |= ((ok >> 15) | (ok >> 30)) << (2 * i); // bit 15 -> 0, bit 31 -> 1 (stray bit 16 masked below)
            }
            bits &= 0xFFu;
            const int first_pos = r * kRowSamples + 8 * lane;
            const int nvalid    = (int)npos - first_pos;
            if (nvalid < 8) bits &= (nvalid <= 0) ? 0u : ((1u << nvalid) - 1u);
            surv |= (uint64_t)bits << (8 * r);
        }

        // ---------------- compact stage-1 survivors into the queue (lane-major order; order is irrelevant)
        uint32_t n1;
        uint32_t at = wave_excl_scan((uint32_t)__builtin_popcountll(surv), lane, &n1);
        while (surv)
        {
            const int b   = __builtin_ctzll(surv);
            surv &= surv - 1;
            queue[at++] = (uint16_t)((b >> 3) * kRowSamples + 8 * lane + (b & 7));
        }
        __syncthreads();

        // ---------------- stage 2 (:794-811), dense over lanes; survivors are compacted in place
        uint32_t n2 = 0;
        for (uint32_t base = 0; base < n1; base += 64)
        {
            const uint32_t idx = base + (uint32_t)lane;
            bool           ok  = false;
            uint32_t       pos = 0;
            if (idx < n1)
            {
                pos          = queue[idx];
                const int w  = kFront + (int)pos;
                const int m0 = a.lut[tile[w]], m2 = a.lut[tile[w + 2]], m7 = a.lut[tile[w + 7]], m9 = a.lut[tile[w + 9]];
                const int high = (m0 + m2 + m7 + m9) / 6;
                ok = (int)a.lut[tile[w + 4]] < high && (int)a.lut[tile[w + 5]] < high && (int)a.lut[tile[w + 11]] < high
                     && (int)a.lut[tile[w + 12]] < high && (int)a.lut[tile[w + 13]] < high && (int)a.lut[tile[w + 14]] < high;
            }
            const uint64_t mk = ballot(ok);
            if (ok) queue[n2 + (uint32_t)__builtin_popcountll(mk & ((1ull << lane) - 1ull))] = (uint16_t)pos;
            n2 += (uint32_t)__builtin_popcountll(mk);
        }
        __syncthreads();

        // ---------------- demodulate the candidates, one at a time, whole wave each
        Emit e;
        e.base   = a.chunk_records + (uint64_t)chunk * a.cap;
        e.cap    = a.cap;
        e.count  = 0;
        e.buffer = bidx;
        for (uint32_t t = 0; t < n2; t++)
        {
            const uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)queue[t]);
            demod_candidate(tile, a.lut, lane, lt, e, kFront + (int)pos, g0 + pos);
        }
        if (lane == 0) a.chunk_counts[chunk] = e.count;
    }
}

// ---------------------------------------------------------------------------------------------
// ordering pass: exclusive prefix over the per-chunk counts (one workgroup), then one wave per chunk copies its
// records into the dense array sorted by (offset, pass).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void prefix_counts_kernel(const uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets,
                                                             uint32_t nchunks, uint32_t cap, uint32_t* __restrict__ total_overflow)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t ovf;
    const uint32_t      tid = threadIdx.x;
    const uint32_t      per = (nchunks + 1023u) / 1024u;
    const uint32_t      lo  = tid * per;
    const uint32_t      hi  = (lo + per < nchunks) ? lo + per : nchunks;
    if (tid == 0) ovf = 0;
    __syncthreads();
    uint32_t sum = 0, over = 0;
    for (uint32_t c = lo; c < hi; c++)
    {
        uint32_t v = counts[c];
        if (v > cap)
        {
            v    = cap;
            over = 1;
        }
        sum += v;
    }
    if (over) atomicOr(&ovf, 1u);
    part[tid] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partial sums
    for (uint32_t o = 1; o < 1024u; o <<= 1)
    {
        uint32_t v = (tid >= o) ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint32_t c = lo; c < hi; c++)
    {
        offsets[c] = run;
        uint32_t v = counts[c];
        run += (v > cap) ? cap : v;
    }
    if (tid == 1023u)
    {
        total_overflow[0] = part[1023];
        total_overflow[1] = ovf;
    }
}

__global__ __launch_bounds__(256) void gather_sorted_kernel(const adsb_amd_record_t* __restrict__ chunk_records,
                                                            const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                            uint32_t nchunks, uint32_t cap, adsb_amd_record_t* __restrict__ dense)
{
    const int      lane   = threadIdx.x & 63;
    const uint32_t wave   = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t c = wave; c < nchunks; c += nwaves)
    {
        uint32_t n = counts[c];
        if (n == 0) continue;
        if (n > cap) n = cap;
        const adsb_amd_record_t* src = chunk_records + (uint64_t)c * cap;
        adsb_amd_record_t*       dst = dense + offsets[c];
        for (uint32_t i = (uint32_t)lane; i < n; i += 64)
        {
            const uint4*   p   = reinterpret_cast<const uint4*>(src + i);
            const uint4    lo  = p[0], hi = p[1];
            const uint32_t key = (lo.y << 1) | ((hi.x >> 8) & 1u); // (offset, pass)
            uint32_t       rank = 0;
            for (uint32_t k = 0; k < n; k++)
            {
                const uint32_t* q  = reinterpret_cast<const uint32_t*>(src + k);
                const uint32_t  kk = (q[1] << 1) | ((q[4] >> 8) & 1u);
                rank += (kk < key) ? 1u : 0u;
            }
            uint4* o = reinterpret_cast<uint4*>(dst + rank);
            o[0]     = lo;
            o[1]     = hi;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// parity helpers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void magnitude1090_kernel(const uint8_t* __restrict__ iq, uint16_t* __restrict__ mag, size_t n,
                                                            const uint16_t* __restrict__ lut)
{
    size_t i      = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) mag[i] = lut[iq1_to_s(iq[2 * i], iq[2 * i + 1])];
}

__global__ __launch_bounds__(256) void phase978_kernel(const uint8_t* __restrict__ iq, uint16_t* __restrict__ phi, size_t n,
                                                       const uint16_t* __restrict__ lut)
{
    size_t i      = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) phi[i] = lut[(uint32_t)iq[2 * i] | ((uint32_t)iq[2 * i + 1] << 8)];
}

} // namespace

hipError_t launch_scan1090(const ScanArgs& a, adsb_amd_record_t* dense, uint32_t* chunk_offsets, uint32_t* total_and_overflow,
                           hipStream_t stream, hipEvent_t ev_scan_begin, hipEvent_t ev_scan_end)
{
    if (a.total_chunks == 0)
    {
        return hipMemsetAsync(total_and_overflow, 0, 2 * sizeof(uint32_t), stream);
    }
    // persistent single-wave workgroups: enough to fill 256 CUs at the LDS-limited occupancy, multiple of 8 (XCDs)
    uint32_t grid = 256u * 12u;
    if (grid > a.total_chunks) grid = ((a.total_chunks + 7u) / 8u) * 8u;
    if (ev_scan_begin) (void)hipEventRecord(ev_scan_begin, stream);
    hipLaunchKernelGGL(scan1090_kernel, dim3(grid), dim3(64), 0, stream, a);
    if (ev_scan_end) (void)hipEventRecord(ev_scan_end, stream);
    hipLaunchKernelGGL(prefix_counts_kernel, dim3(1), dim3(1024), 0, stream, a.chunk_counts, chunk_offsets, a.total_chunks, a.cap,
                       total_and_overflow);
    uint32_t gblocks = (a.total_chunks + 3u) / 4u;
    if (gblocks > 2048u) gblocks = 2048u;
    hipLaunchKernelGGL(gather_sorted_kernel, dim3(gblocks), dim3(256), 0, stream, a.chunk_records, a.chunk_counts, chunk_offsets,
                       a.total_chunks, a.cap, dense);
    return hipGetLastError();
}

hipError_t launch_magnitude1090(const uint8_t* iq, uint16_t* mag, size_t nsamples, const uint16_t* lut, hipStream_t stream)
{
    if (nsamples == 0) return hipSuccess;
    size_t blocks = (nsamples + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(magnitude1090_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, iq, mag, nsamples, lut);
    return hipGetLastError();
}

hipError_t launch_phase978(const uint8_t* iq, uint16_t* phi, size_t nsamples, const uint16_t* lut65536, hipStream_t stream)
{
    if (nsamples == 0) return hipSuccess;
    size_t blocks = (nsamples + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(phase978_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, iq, phi, nsamples, lut65536);
    return hipGetLastError();
}

// magnitude of s = i*i + q*q exactly as the reference LUT defines it (ADSB1090.cpp:138): round(sqrt(s) * 360).
// Integer form: m is the integer with m*m - m < 129600*s <= m*m + m (ties cannot occur), checked against the
// double-precision formula in the tests.  Index 32767 aliases s = 32768 (see iq2_to_s2).
void build_mag_lut(uint16_t* lut)
{
    for (uint32_t idx = 0; idx < (uint32_t)kLutSize; idx++)
    {
        uint64_t s = (idx == 32767u) ? 32768u : idx;
        if (s == 0)
        {
            lut[idx] = 0;
            continue;
        }
        uint64_t t = 129600ull * s;
        uint64_t m = (uint64_t)(sqrt((double)s) * 360.0 + 0.5);
        while (m * m - m >= t) m--;
        while (t > m * m + m) m++;
        lut[idx] = (uint16_t)m;
    }
}

void build_crc_table(uint32_t* tab)
{
    uint32_t t = 0xFFF409u; // x^24 mod G, G = x^24 + 0xFFF409
    for (int j = 87; j >= 0; j--)
    {
        tab[j] = t;
        t <<= 1;
        if (t & 0x1000000u) t ^= 0x1FFF409u;
    }
    for (int j = 88; j < 112; j++) tab[j] = 0;
}

} // namespace adsb_amd
