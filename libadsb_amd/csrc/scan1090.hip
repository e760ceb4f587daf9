// scan1090.hip -- gfx950 (MI355X, CDNA4) kernels for the 1090ES IQ -> Mode S candidate-record path.
//
// One wavefront (a 64-thread workgroup) owns one "chunk": 4096 consecutive preamble positions of one
// reference buffer.  It streams the 8.5 KiB of u8 IQ that those positions can touch with 16-byte
// coalesced loads (the next chunk's loads are issued before the current chunk is processed, so HBM
// latency hides behind the arithmetic), turns every I,Q pair into s = (I-127)^2 + (Q-127)^2 with packed
// 16-bit math and parks s (u16) in a wave-private LDS image with the two halves of the chunk interleaved
// (scan1090.h): dword q = (s[q], s[q + 2048]).  Everything after that works out of LDS:
//
//   stage 1  (reference ADSB1090.cpp:782-783)  ten preamble comparisons per position, evaluated on s
//            instead of on the magnitude: the reference LUT m = round(360*sqrt(s)) is strictly
//            increasing on every reachable s, so <,> between magnitudes equal <,> between s values.
//            Packed: one VALU op works on two positions 2048 samples apart, operand "sample q+a" is dword q+a
//            for every a, the ten comparisons fold into four "low < high" bounds with shared sliding maxima.
//   stage 2  (:794-811) exact magnitudes for the ~1 % of positions that survive, dense over lanes.
//   demod    (:814-881, 277-332) one candidate at a time, all 64 lanes: lane b slices bit b (and 64+b),
//            the "copy the previous bit" and the phase-correction recurrences are resolved from
//            __ballot masks, parity is a DPP XOR-reduction of per-lane table entries, the 1-bit repair
//            is a ballot over per-lane syndromes.
//
// Magnitudes are computed, not looked up: m = round(360*sqrt(s)) is the integer with
// m*m - m < 129600*s <= m*m + m, so a float estimate plus one exact integer correction reproduces the
// reference's double-precision LUT bit for bit (checked over every I,Q pair in the tests).
//
// The sequential part of the reference (skip-ahead after an accepted frame, ICAO-cache gating of
// AP-type DFs; :886-957) is NOT done here: the kernel emits one record per (offset, pass) that the
// reference could accept and the host resolver applies those rules in sample order.
//
// HBM traffic: the IQ bytes once (+6 % halo, normally an L2 hit: neighbouring chunks are scheduled on
// the same XCD) and 32 B per record.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <math.h>
#include <stdint.h>

#include "decode1090.h"
#include "scan1090.h"
#include "gather1090.hip.h"
#include "scan_common.hip.h"

namespace adsb_amd
{
namespace
{

constexpr int kQueueCap  = 256;                               // stage-1 survivors processed per pass (a chunk rarely has more than ~60)
constexpr int kLdsDwords = kTileDwords + kQueueCap / 2 + 12;   // 2448 dwords = 9792 bytes per wave (16 waves per CU fit in 160 KiB)
static_assert(kLdsDwords > (kHalfChunk - 1) + 2 * 63 + 145, "reads of the fast demodulation path (load_bit_pairs) must stay inside the array");

// Parity / repair / classification of one sliced message (wave-uniform ba/bb = message bits 0..63 / 64..111,
// LSB = lowest bit index).  *stateless is set when the frame is one the reference accepts without consulting
// the ICAO cache (DF11/17 with good or repaired parity).
__device__ __forceinline__ void classify_and_emit(Emit& e, int lane, const LaneTables& lt, uint64_t ba, uint64_t bb, uint32_t df,
                                                  uint32_t nbits, uint32_t offset, uint32_t flags, bool* stateless)
{
    *stateless      = false;
    const bool is17 = (df == 11 || df == 17);
    if (!is17 && !df_is_ap(df)) return; // BruteForceAp returns 0 for every other DF (:403-409)

    if (nbits == 56)
    { // a short frame keeps only its 56 bits: what the slicer produced beyond them is noise nothing downstream reads
        ba &= kMask56;
        bb = 0;
    }
    const bool bit_a = (ba >> lane) & 1ull;
    const bool bit_b = (lane < 48) && ((bb >> lane) & 1ull);
    uint32_t   contrib, stored;
    if (nbits == 112)
    {
        contrib = (bit_a ? lt.crc_a : 0u) ^ (bit_b ? lt.crc_b : 0u);
        stored  = (uint32_t)(__builtin_bitreverse64(bb) >> 16) & 0xFFFFFFu; // message bits 88..111, first bit = MSB
    }
    else
    {
        contrib = (bit_a && lane < 56) ? lt.crc_s : 0u;
        stored  = (uint32_t)(__builtin_bitreverse64(ba) >> 8) & 0xFFFFFFu; // message bits 32..55
    }
    const uint32_t syn = wave_xor(contrib) ^ stored;
    if (is17)
    {
        int errorbit = -1;
        if (syn != 0)
        {
            // first bit (ascending) whose flip makes stored == computed
            if (nbits == 112)
            {
                int ol = lane; // (a lane number the compiler cannot see through: the two syndromes below are not hoisted out of the scan loop)
                asm volatile("" : "+v"(ol));
                uint64_t ma = ballot(syn == lt.crc_a);
                uint64_t mb = ballot(ol < 48 && syn == lt.syn_b(ol));
                if (ma) errorbit = __builtin_ctzll(ma);
                else if (mb) errorbit = 64 + __builtin_ctzll(mb);
            }
            else
            {
                int ol = lane;
                asm volatile("" : "+v"(ol));
                uint64_t ms = ballot(ol < 56 && syn == lt.syn_s(ol));
                if (ms) errorbit = __builtin_ctzll(ms);
            }
            if (errorbit < 0) return;
        }
        *stateless = true;
        emit_raw(e, lane, offset, ba, bb, df, nbits, errorbit, flags, 0u);
        return;
    }
    // AP-type: address candidate = AP xor parity (:418-425); validity is decided against the ICAO cache on the host
    emit_raw(e, lane, offset, ba, bb, df, nbits, -1, flags | ADSB_AMD_F_NEEDS_ICAO, syn);
}

// Exact per-lane slicing inputs of one candidate (reference magnitudes of the two samples of bit `lane` and bit 64+lane).
struct BitMags
{
    int loA, hiA, loB, hiB;
};

// Where one candidate's samples live in the interleaved image (scan1090.h): sample t = 0 .. 239 of the window of chunk position
// pos is the half at a0 + 2 t.  All members are wave-uniform.
struct Win
{
    uint32_t a0;  // tile index (uint16_t units) of sample t = 0
    uint32_t am1; // tile index of sample t = -1
};

// s of the two samples of bit `lane` (t = 16 + 2 lane, +1) and of bit 64 + lane (another 128 samples on; lanes >= 48 repeat bit `lane`)
__device__ __forceinline__ void load_bit_samples(const uint16_t* tile, const Win& w, int lane, bool has_b, uint32_t& sLoA, uint32_t& sHiA,
                                                 uint32_t& sLoB, uint32_t& sHiB)
{
    const uint32_t ia = w.a0 + 32u + 4u * (uint32_t)lane, ib = has_b ? ia + 256u : ia;
    sLoA = tile[ia]; sHiA = tile[ia + 2]; sLoB = tile[ib]; sHiB = tile[ib + 2];
}

// DetectOutOfPhase (:683-690) != 0 for the preamble of window w (j >= 1): needs exact magnitudes of m[j-1 .. j+10].  The sample in
// front of the chunk's first position is not in the image: it is fetched here, from the buffer, the one time in 4096 it is needed
// (`front` = its address; computing it per chunk cost seven vector instructions).
__device__ __forceinline__ bool preamble_out_of_phase(const uint16_t* tile, int lane, const Win& w, const uint8_t* front)
{
    const uint32_t t = (uint32_t)(lane & 15); // lanes 0..15: m[j-1 .. j+14]
    uint32_t       s = tile[t == 0 ? w.am1 : w.a0 + 2u * (t - 1u)];
    if (w.a0 == 0)
    {
        const uint32_t f = *reinterpret_cast<const uint16_t*>(front);
        if (t == 0) s = iq1_to_s(f & 0xFFu, f >> 8);
    }
    const int pre = mag_of_s(s);
    const int m_1 = __builtin_amdgcn_readlane(pre, 0), m1 = __builtin_amdgcn_readlane(pre, 2), m2 = __builtin_amdgcn_readlane(pre, 3);
    const int m3 = __builtin_amdgcn_readlane(pre, 4), m6 = __builtin_amdgcn_readlane(pre, 7), m7 = __builtin_amdgcn_readlane(pre, 8);
    const int m9 = __builtin_amdgcn_readlane(pre, 10), m10 = __builtin_amdgcn_readlane(pre, 11);
    // x > y/3 (integer division)  <=>  3x > y
    return (3 * m3 > m2) || (3 * m10 > m9) || (3 * m6 > m7) || (3 * m_1 > m1);
}

// LDS byte address of a __shared__ object
__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }

// The s values of a lane's two bits as packed pairs: lo = (s of the first sample of bit `lane` | s of the first sample of bit 64 + lane << 16),
// hi the same for the second samples.  A window's samples are the same half (lower: positions < 2048, upper: the rest) of consecutive
// dwords of the image, so the four values are two aligned dword pairs (one ds_read2_b32 each) and one byte permute per result picks
// the window's half of both.  `a0` = index of sample t = 0 of the window in halves (wave-uniform).  Lanes >= 48 have no second bit:
// they read whatever follows the window (the array is long enough) and every use of their upper halves is masked.
// (ds_read_u16_d16_hi into a register that already holds the other half would save the permutes, but with SRAM ECC on, as on this
// part, a d16 load does not keep the other half: tools/isa_probe.hip.)
__device__ __forceinline__ void load_bit_pairs(uint32_t tile_addr, uint32_t a0, int lane, uint32_t& lo, uint32_t& hi)
{
    const uint32_t addr = tile_addr + 4u * (a0 >> 1) + 8u * (uint32_t)lane;
    const uint32_t sel  = (a0 & 1u) ? 0x07060302u : 0x05040100u;
    uint64_t       xa, xb;
    asm volatile("ds_read2_b32 %0, %2 offset0:16 offset1:17\n\t"
                 "ds_read2_b32 %1, %2 offset0:144 offset1:145\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(xa), "=&v"(xb)
                 : "v"(addr)
                 : "memory");
    lo = __builtin_amdgcn_perm((uint32_t)xb, (uint32_t)xa, sel);
    hi = __builtin_amdgcn_perm((uint32_t)(xb >> 32), (uint32_t)(xa >> 32), sel);
}

// Ballot of the sign bit of the lower / upper half of a packed pair, and "mask bit of this lane ? v : 0": one vector instruction each
// (written out because the compiler turns a test of bit 15 into an AND or a bit-field extract followed by a compare).
__device__ __forceinline__ uint64_t sign16_ballot(uint32_t x)
{
    uint64_t m;
    asm("v_cmp_lt_i16_e64 %0, %1, 0" : "=s"(m) : "v"(x));
    return m;
}
__device__ __forceinline__ uint64_t sign32_ballot(uint32_t x)
{
    uint64_t m;
    asm("v_cmp_lt_i32_e64 %0, %1, 0" : "=s"(m) : "v"(x));
    return m;
}
__device__ __forceinline__ uint32_t select_by_mask(uint64_t mask, uint32_t v)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(mask));
    return r;
}

// The common cases, nearly straight-line and in packed integer arithmetic on s only (a lane's two bits share every instruction).
//
// (1) A frame whose every relevant bit is far above the "decided" and energy thresholds.  A bit is called strong when the larger
//     of its two samples has L >= 2 S + 108 against the smaller one: then sqrt(L) - sqrt(S) >= sqrt(2 S + 108) - sqrt(S) >= 7.348
//     (minimum at S = 54), so the reference's magnitudes differ by at least 360 * 7.348 - 1 = 2644, far above 256 (the bit is
//     decided, :838) and above 2550 (if every bit of the frame is strong the energy average passes, :870-877).  The sliced bits
//     are then simply "which half is larger" (an exact comparison of s) and pass 1 is settled here: DF11/17 with good parity or a
//     single repairable bit is accepted by the reference on the spot; an AP-type DF yields its conditional record and, unless
//     the preamble is out of phase, the retry would reproduce the same bits.
//     In 16-bit arithmetic: t = min(2 S + 107, 32767) (signed saturation), strong <=> L > t.  When 2 S + 107 saturates no L
//     exceeds it, which is right: L <= 32767 < 2 S + 108.  (32767 standing for s = 32768 only makes the test stricter.)
// (2) Noise that got through the preamble gates (40 % of the candidates of a quiet band).  |lo - hi| <= max(lo, hi) =
//     round(360 sqrt(max s)) <= 360 sqrt(max s) + 1/2, and by Cauchy-Schwarz sum_b sqrt(x_b) <= sqrt(n sum_b x_b); so with
//     S_n = sum over the first n bits of max(s_lo, s_hi):  sum |lo - hi| <= 360 sqrt(n S_n) + n/2.  The gate needs
//     sum |lo - hi| >= 1275 n (:870-877 with msglen * 4 = n / 2), impossible once S_n < n (1274.5 / 360)^2 = 12.5336 n, i.e.
//     S_56 <= 701 resp. S_112 <= 1403.  If that holds for BOTH lengths the candidate is dead whichever DF its bits spell, on
//     both passes (the retry restores the window before the gate, :855-856).  Every term is clipped at 1023 first so that the two
//     sums fit the two halves of one register (56 x 1023 < 65536): a clipped term alone exceeds both bounds.
// Returns 0 when the candidate is finished, 1 when the general demodulator has to run from scratch (nothing was
// emitted), 2 when only its retry pass remains (the pass-1 record is already out).
struct FastConsts
{
    uint32_t two;      // (2, 2)
    uint32_t sel_sums; // per lane: v_perm selector that routes a lane's two maxima into the (S_56 | rest) halves
};
__device__ __forceinline__ FastConsts fast_consts(int lane)
{
    FastConsts c;
    c.two      = 0x00020002u;
    // bits 0..55 count for S_56 (low half), bits 56..111 for the rest (high half): lanes 0..47 hold one of each, lanes 48..55 a
    // first-half bit only, lanes 56..63 a second-half bit in their LOWER half
    c.sel_sums = lane < 48 ? 0x03020100u : (lane < 56 ? 0x0C0C0100u : 0x01000C0Cu);
    return c;
}
__device__ __forceinline__ int demod_strong_frame(const uint16_t* tile, uint32_t tile_addr, int lane, const LaneTables& lt, const FastConsts& fc, Emit& e,
                                                  const Win& w, uint32_t j, const uint8_t* front)
{
    uint32_t lo, hi;
    load_bit_pairs(tile_addr, w.a0, lane, lo, hi);
    const uint32_t d    = pk_sub(hi, lo);                // sign set <=> first sample larger <=> bit = 1 (all values < 32768)
    const uint64_t valA = sign16_ballot(d), valB32 = sign32_ballot(d), valB = valB32 & kMask48;
    const uint32_t mx   = pk_max(lo, hi), mn = as_u32(__builtin_elementwise_min(as_pk(lo), as_pk(hi)));
    uint32_t       t;
    asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(t) : "v"(mn), "v"(fc.two), "s"(0x006B006Bu)); // min(2 S + 107, 32767)
    const uint32_t st      = pk_sub(t, mx);              // sign set <=> L > t <=> strong
    const uint64_t strongA = sign16_ballot(st), strongB = sign32_ballot(st);
    const uint32_t df      = (uint32_t)(__builtin_bitreverse64(valA) >> 59);
    const bool     is_long = df_is_long(df);
    const bool     is17    = (df == 17 || df == 11);
    const bool     strong  = is_long ? (strongA == ~0ull && (strongB & kMask48) == kMask48) : ((strongA & kMask56) == kMask56);
    if (!strong || !(is17 || df_is_ap(df)))
    {
        const u16x2    lim = {1023, 1023};
        const uint32_t trm = as_u32(__builtin_elementwise_min(as_pk(__builtin_amdgcn_perm(0u, mx, fc.sel_sums)), lim));
        const uint32_t sum = wave_sum(trm);
        const uint32_t s56 = sum & 0xFFFFu, s112 = s56 + (sum >> 16);
        return (int)((701u - s56) >> 31) | (int)((1403u - s112) >> 31); // 0: both within the bounds (the sums are < 2^17)
    }
    const uint64_t ba = is_long ? valA : (valA & kMask56);
    const uint64_t bb = is_long ? valB : 0ull;
    const uint32_t nbits = is_long ? 112u : 56u;
    // crc_b is 0 on lanes >= 48, crc_s on lanes >= 56
    const uint32_t contrib = is_long ? (select_by_mask(valA, lt.crc_a) ^ select_by_mask(valB32, lt.crc_b)) : select_by_mask(valA, lt.crc_s);
    const uint32_t stored  = (is_long ? (uint32_t)(__builtin_bitreverse64(bb) >> 16) : (uint32_t)(__builtin_bitreverse64(ba) >> 8)) & 0xFFFFFFu;
    const uint32_t syn = wave_xor(contrib) ^ stored;
    if (is17)
    {
        int errorbit = -1;
        if (syn != 0)
        { // FixSingleBitErrors (:304-332): first bit whose flip makes stored == computed
            int ol = lane; // (a lane number the compiler cannot see through: the two syndromes are worked out here, not kept across the scan loop)
            asm volatile("" : "+v"(ol));
            const uint64_t ma = is_long ? ballot(syn == lt.crc_a) : ballot(ol < 56 && syn == lt.syn_s(ol));
            const uint64_t mb = is_long ? ballot(ol < 48 && syn == lt.syn_b(ol)) : 0ull;
            if (ma) errorbit = __builtin_ctzll(ma);
            else if (mb) errorbit = 64 + __builtin_ctzll(mb);
            else return 1; // not repairable: the reference retries with phase correction
        }
        emit_raw(e, lane, j, ba, bb, df, nbits, errorbit, 0u, 0u);
        return 0;
    }
    emit_raw(e, lane, j, ba, bb, df, nbits, -1, ADSB_AMD_F_NEEDS_ICAO, syn);
    if (j == 0 || !preamble_out_of_phase(tile, lane, w, front)) return 0; // the retry would slice the same window again
    // The retry (:814-826, ApplyPhaseCorrection :720-736) turns the first sample x of every bit after the first into (x * 5) / 4 or
    // (x * 4) / 5, 16 bits wide.  On a frame whose bits are all strong that changes nothing as long as no product wraps:
    //   bit 1 (L first):  (4 L') / 5 - H' >= 288 sqrt(2 S + 108) - 360 sqrt(S) - 1.9 >= 1400  (minimum at S = 193), and the larger factor
    //                     only widens the gap;
    //   bit 0 (L second): H' - (5 S') / 4 >= 360 sqrt(2 S + 108) - 450 sqrt(S) - 2.1 >= 1750, the smaller factor widens it;
    // (primes: magnitudes, each within 1/2 of 360 sqrt(s)), both far above the 256 of "decided" (:838), whichever factor the chain
    // picks.  The window is restored before the energy gate (:855-856), which therefore passes as it did in pass 1.  So the retry spells the
    // same message and yields the same conditional record, now flagged PASS2 | PHASE -- unless (x * 5) / 4 can exceed 65535, i.e. some
    // magnitude is above 52428, i.e. s > 21209: then the general demodulator runs the retry (wrapped samples decide differently).
    const uint32_t big   = pk_sub(0x52085208u, mx); // 21000 - max(s): sign set <=> above
    const uint64_t wrapA = sign16_ballot(big), wrapB = is_long ? (sign32_ballot(big) & kMask48) : 0ull;
    if ((wrapA | wrapB) != 0) return 2;
    emit_raw(e, lane, j, ba, bb, df, nbits, -1, ADSB_AMD_F_NEEDS_ICAO | ADSB_AMD_F_PASS2 | ADSB_AMD_F_PHASE, syn);
    return 0;
}

// Demodulate the candidate of window w (preamble at sample j of the buffer).
//
// Pass 1 is first attempted on float magnitude estimates: which half of a bit is larger is an exact comparison of s
// (the magnitude is strictly increasing in s), "|lo-hi| >= 256" (:838) and the energy gate (:870-877) are decided from
// the estimates whenever they are further from their thresholds than the estimate's error bound.  Only when some
// decision is inside that margin -- or the retry slice, which rescales exact magnitudes, is needed -- are the exact
// magnitudes computed.  Either way the bits that come out are exactly the reference's.
__device__ __forceinline__ void demod_candidate(const uint16_t* tile, int lane, const LaneTables& lt, Emit& e, const Win& w, uint32_t j, bool pass1_done,
                                                const uint8_t* front)
{
    const bool has_b = lane < 48;
    // bit `lane` lives in samples j+16+2*lane, j+17+2*lane; bit 64+lane another 128 samples on (ADSB1090.cpp:831-835)
    uint32_t sLoA, sHiA, sLoB, sHiB;
    load_bit_samples(tile, w, lane, has_b, sLoA, sHiA, sLoB, sHiB);

    // bit 0 with equal halves is the reference's only reachable "errors++" (:839-846); it survives the retry
    // unchanged (sample j+16 is never rescaled), so such a candidate can never be decoded.  m equal <=> s equal.
    if (__builtin_amdgcn_readfirstlane((int)(sLoA == sHiA))) return;

    const bool valA = sLoA > sHiA, valB = has_b && (sLoB > sHiB);
    BitMags    x{};
    bool       have_exact = false;
    uint32_t   sum56 = 0, sumrest = 0;
    uint64_t   ba = 0, bb = 0;
    uint32_t   df = 0, nbits = 0;
    bool       stateless = false;
    // exact magnitudes of the first 64 bits' samples: the prelude below needs five of them, and whoever needs exact values later has these
    x.loA = mag_of_s(sLoA), x.hiA = mag_of_s(sHiA);
    if (!pass1_done)
    {
        // ---------------- prelude (round 5): can this candidate yield a record at all?  A record needs a DF the reference can accept (11 / 17, or an
        // AP-type one) on pass 1 or on the retry, and the DF is the first five sliced bits.  Those are worked out here exactly, on the scalar unit,
        // from ten magnitudes -- pass 1 (:830-853: bit 0 by comparison, a later bit copies its predecessor when |lo - hi| < 256), then, if pass 1
        // cannot be accepted, the retry: none unless the preamble is out of phase (:814-826 -- then the retry slices the same window again), else
        // ApplyPhaseCorrection's chain over bits 0 .. 4 (:720-736: the first sample of bit k + 1 becomes (x 5) / 4 or (x 4) / 5, 16 bits wide, by
        // whether the already rescaled first sample of bit k exceeds its second).  Neither DF acceptable: nothing can come of the candidate
        // whatever its energy gate says (:870-881 only decides between "dead" and "retry"), and the estimates, both energy sums, the 112-bit
        // slicing, parity and the whole retry are skipped.  At noise +-20 that is one candidate in five (profiles/r05_sensitivity.txt).
        int lo5[5], hi5[5];
#pragma unroll
        for (int k = 0; k < 5; k++) lo5[k] = __builtin_amdgcn_readlane(x.loA, k), hi5[k] = __builtin_amdgcn_readlane(x.hiA, k);
        uint32_t df1 = 0, prev = 0;
#pragma unroll
        for (int k = 0; k < 5; k++)
        {
            const bool     decided = k == 0 || abs(lo5[k] - hi5[k]) >= 256;
            const uint32_t bit     = decided ? (uint32_t)(lo5[k] > hi5[k]) : prev;
            df1 = (df1 << 1) | bit, prev = bit;
        }
        if (!(df1 == 11u || df1 == 17u || df_is_ap(df1)))
        {
            if (j == 0 || !preamble_out_of_phase(tile, lane, w, front)) return; // the retry would spell the same DF
            uint32_t df2 = 0;
            int      lo2 = lo5[0]; // the (rescaled) first sample of the bit at hand
            prev         = 0;
#pragma unroll
            for (int k = 0; k < 5; k++)
            {
                const bool     first_larger = lo2 > hi5[k];
                const bool     decided      = k == 0 || abs(lo2 - hi5[k]) >= 256;
                const uint32_t bit          = decided ? (uint32_t)first_larger : prev;
                df2 = (df2 << 1) | bit, prev = bit;
                if (k < 4) lo2 = first_larger ? (int)(uint16_t)((lo5[k + 1] * 5) / 4) : (int)(uint16_t)((lo5[k + 1] * 4) / 5);
            }
            if (!(df2 == 11u || df2 == 17u || df_is_ap(df2))) return;
        }
        // ---------------- pass 1 on estimates
        const float fA = (float)abs(x.loA - x.hiA); // (exact since the prelude: never inside the margins below)
        const float fB = has_b ? __builtin_fabsf(mag_estimate(sLoB) - mag_estimate(sHiB)) : 0.0f;
        const float lo_edge = 256.0f - 2.0f * kEstErr, hi_edge = 256.0f + 2.0f * kEstErr;
        bool        need_exact = false;
        // Strong clean frame: every one of the 112 bits has |lo-hi| far above both the "decided" threshold and the
        // energy-gate average, so the value ballots are the message and the gate passes for either length.
        if (ballot(fA >= 2560.0f) == ~0ull && ballot(has_b && fB >= 2560.0f) == kMask48)
        {
            ba    = ballot(valA);
            bb    = ballot(valB);
            df    = (uint32_t)(__builtin_bitreverse64(ba) >> 59);
            nbits = df_is_long(df) ? 112u : 56u;
        }
        else
        {
            const bool unsure = has_b && fB > lo_edge && fB < hi_edge; // (the first 64 bits' differences are exact)
            need_exact        = ballot(unsure) != 0;
            if (!need_exact)
            {
                const uint32_t iA = (uint32_t)(fA + 0.5f), iB = (uint32_t)(fB + 0.5f);
                uint32_t e56, erest;
                wave_sum2(lane < 56 ? iA : 0u, (lane >= 56 ? iA : 0u) + iB, e56, erest);
                slice_resolve(lane, has_b, lane == 0 || fA >= 256.0f, valA, fB >= hi_edge, valB, &ba, &bb);
                df    = (uint32_t)(__builtin_bitreverse64(ba) >> 59);
                nbits = df_is_long(df) ? 112u : 56u;
                // each |lo-hi| estimate is within 2*kEstErr + 0.5 of the true integer, so is the average
                const uint32_t avg = (nbits == 112u) ? (e56 + erest) / 56u : e56 / 28u;
                if (avg + 5u < 2550u) return;        // surely below the gate (:877-881): dead, no retry
                if (avg < 2550u + 5u) need_exact = true; // too close to call
            }
        }
        if (need_exact)
        {
            x.loB = mag_of_s(sLoB); x.hiB = mag_of_s(sHiB);
            have_exact   = true;
            const int dA = abs(x.loA - x.hiA), dB = has_b ? abs(x.loB - x.hiB) : 0;
            wave_sum2(lane < 56 ? (uint32_t)dA : 0u, (lane >= 56 ? (uint32_t)dA : 0u) + (uint32_t)dB, sum56, sumrest);
            slice_resolve(lane, has_b, lane == 0 || dA >= 256, valA, dB >= 256, valB, &ba, &bb);
            df    = (uint32_t)(__builtin_bitreverse64(ba) >> 59);
            nbits = df_is_long(df) ? 112u : 56u;
            const uint32_t delta = (nbits == 112u) ? (sum56 + sumrest) / 56u : sum56 / 28u;
            if (delta < 2550u) return; // :877-881, no retry either
        }
        classify_and_emit(e, lane, lt, ba, bb, df, nbits, j, 0u, &stateless);
        if (stateless) return; // the reference accepts here and never retries
        // ---------------- pass 2: retry with phase correction (:814-826); identical to pass 1 unless the window is rescaled
        if (j == 0 || !preamble_out_of_phase(tile, lane, w, front)) return;
    }
    if (!have_exact)
    {
        x.loB = mag_of_s(sLoB); x.hiB = mag_of_s(sHiB);
        const int dA = abs(x.loA - x.hiA), dB = has_b ? abs(x.loB - x.hiB) : 0;
        wave_sum2(lane < 56 ? (uint32_t)dA : 0u, (lane >= 56 ? (uint32_t)dA : 0u) + (uint32_t)dB, sum56, sumrest);
    }
    const int loA = x.loA, hiA = x.hiA, loB = x.loB, hiB = x.hiB;

    // ApplyPhaseCorrection (:720-736): the first sample of bit k>=1 becomes up(x) or dn(x) (u16 wrap) depending on
    // whether the (already rescaled) first sample of bit k-1 exceeds its second sample.
    const int  upA = (int)(uint16_t)((loA * 5) / 4), dnA = (int)(uint16_t)((loA * 4) / 5);
    const int  upB = (int)(uint16_t)((loB * 5) / 4), dnB = (int)(uint16_t)((loB * 4) / 5);
    const bool cuA = (lane == 0) ? (loA > hiA) : (upA > hiA);
    const bool cdA = (lane == 0) ? (loA > hiA) : (dnA > hiA);
    const bool cuB = has_b && (upB > hiB);
    const bool cdB = has_b && (dnB > hiB);
    const uint64_t cstA = ballot(cuA == cdA), vlA = ballot(cuA);
    const uint64_t cstB = ballot(has_b && cuB == cdB), vlB = ballot(cuB);
    uint64_t       TA, TB;
    uint32_t       t63;
    if (cstA == ~0ull && cstB == kMask48)
    { // every step's outcome is independent of the incoming state: the chain is just the value ballots
        TA  = vlA;
        TB  = vlB;
        t63 = (uint32_t)(vlA >> 63);
    }
    else
    {
        const uint64_t ivA = ballot(!cuA && cdA), ivB = ballot(has_b && !cuB && cdB);
        const uint32_t tA  = chain_resolve(cstA, vlA, ivA, lane, 0u);
        t63                = chain_resolve(cstA, vlA, ivA, 63, 0u);
        const uint32_t tB  = chain_resolve(cstB, vlB, ivB, lane, t63);
        TA                 = ballot(tA != 0);
        TB                 = ballot(has_b && tB != 0);
    }
    const uint32_t prevA = (lane == 0) ? 0u : (uint32_t)((TA >> (lane - 1)) & 1ull);
    const uint32_t prevB = (lane == 0) ? t63 : (uint32_t)((TB >> (lane - 1)) & 1ull);
    const int      lo2A  = (lane == 0) ? loA : (prevA ? upA : dnA);
    const int      lo2B  = prevB ? upB : dnB;
    const int      d2A   = abs(lo2A - hiA);
    const int      d2B   = has_b ? abs(lo2B - hiB) : 0;
    slice_resolve(lane, has_b, lane == 0 || d2A >= 256, lo2A > hiA, d2B >= 256, lo2B > hiB, &ba, &bb);
    df    = (uint32_t)(__builtin_bitreverse64(ba) >> 59);
    nbits = df_is_long(df) ? 112u : 56u;
    const uint32_t delta2 = (nbits == 112u) ? (sum56 + sumrest) / 56u : sum56 / 28u; // window is restored before the gate (:855-856)
    if (delta2 < 2550u) return;
    classify_and_emit(e, lane, lt, ba, bb, df, nbits, j, ADSB_AMD_F_PASS2 | ADSB_AMD_F_PHASE, &stateless);
}

// Survivors of stage 1 -> queue.  `surv` bit n stands for the half at index 1024 (n >> 4) + 16 lane + (n & 15) of the image; a queue
// entry is that index (12 bits).  `first` = survivors in lower lanes, `base` = first survivor of this pass of the queue.  The two
// 32-bit halves of the mask are walked one after the other: find-first-set, clear-lowest and the index arithmetic stay 32-bit
// operations (the 64-bit forms cost twice as much), and with the usual ~40 survivors per chunk a lane seldom has more than two in
// either half.  GUARD: more survivors than the queue holds (several passes), every store is checked against the pass's range.
template <bool GUARD>
__device__ __forceinline__ void queue_survivors(uint64_t surv, uint32_t first, int lane, uint32_t queue_addr, uint32_t base)
{
    uint32_t       slot     = (first - base) * 2u + queue_addr; // LDS byte address of this lane's next entry
    const uint32_t end_addr = queue_addr + 2u * (uint32_t)kQueueCap;
#pragma unroll
    for (int h = 0; h < 2; h++)
    {
        uint32_t       sv  = (uint32_t)(surv >> (32 * h));
        const uint32_t org = 2048u * (uint32_t)h + 16u * (uint32_t)lane; // index of bit 0 of this half
        while (sv)
        {
            const uint32_t n = (uint32_t)__builtin_ctz(sv);
            sv &= sv - 1u;
            const uint32_t ti = __umul24(n & 16u, 63u) + (n + org); // 1024 (n >> 4) + (n & 15) + org
            if (!GUARD || (slot >= queue_addr && slot < end_addr)) asm volatile("ds_write_b16 %0, %1" ::"v"(slot), "v"(ti) : "memory");
            slot += 2u;
        }
    }
}

// Measurement builds (diag.hip.h, tools/parts.sh) compile the later parts out: 0 = the loads alone, 1 = + window -> s, 2 = + stage 1,
// 3 = + survivor queue and stage 2, more = everything (the product).
constexpr int kParts = diag::kParts;
__global__ __launch_bounds__(64, 4) void scan1090_kernel(ScanArgs a, uint32_t* __restrict__ total_overflow, GatherArgs ga)
{
    // the interleaved image of s (scan1090.h), the slot of the sample in front of the chunk, then the survivor queue.  The queue
    // doubles as the landing zone of the fast demodulation path's reads beyond a window (lanes 48..63 have no second bit; what
    // they fetch is never used): the last window starts at half 4095 and those reads reach half 4895 < 2 * kLdsDwords.
    __shared__ __attribute__((aligned(16))) uint32_t tile32[kLdsDwords];
    uint16_t* const                                  tile  = reinterpret_cast<uint16_t*>(tile32);
    uint16_t* const                                  queue = reinterpret_cast<uint16_t*>(&tile32[kTileDwords]);

    const int        lane = threadIdx.x;
    stamp(a.stamps, 0);
    const LaneTables lt   = load_lane_tables(a.crc_tab, lane);
    const FastConsts fc   = fast_consts(lane);
    const uint32_t   tile_addr = lds_address(tile32), queue_addr = lds_address(queue);
    if (blockIdx.x == 0) gather_state_zero(total_overflow, (uint32_t)lane); // this slot's ordering state (GatherArgs::state): its pass runs in a later kernel on the stream
    // Round 6: the ordering pass of the OTHER slot's scan -- an earlier kernel on this stream -- by a quarter of this kernel's waves, a block each, before their
    // own chunks (gather1090.hip.h).  The image area is free until then.
    gather_in_front(ga, a.ncu, *reinterpret_cast<GatherLds*>(tile32), (uint32_t)lane);

    // XCD-aware chunk order, one work counter per sub-range: scan_common.hip.h (WorkRange).  (Round 1 also offset the waves of a
    // SIMD in time at the start so that their phases would not coincide; with the priority hint below that costs 1.7 %.)
    WorkRange wr = work_range(a);
    if (wr.slot >= wr.end) return;
    // work items by their chunk number; a wave's first two are fixed, the rest come from its counter (later from others': take_next)
    uint32_t chunk = wr.chunk_of(wr.slot);
    uint32_t next  = wr.slot + wr.nslot < wr.end ? wr.chunk_of(wr.slot + wr.nslot) : kNoChunk;
    Pending  pend{};                          // the previous chunk's directory entry and sums, not yet written (scan_common.hip.h)
    ChunkGeom g     = chunk_geom_of(a, chunk, kFrameSpan);
    RawWindow raw;
    load_window<kHalo, false, true>(g, lane, raw);

    for (;;)
    {
        // ---------------- s = (I-127)^2 + (Q-127)^2 for the window, parked in LDS: row j (lower half) beside row j + 4 (upper half)
        __builtin_amdgcn_s_setprio(0);
        wave_lds_fence(); // readers of the previous chunk are done
        if constexpr (kParts == 0)
        { // measurement build: the loads and nothing else (what does this access pattern alone cost?)
            uint32_t x = raw.cont_hi.x ^ raw.cont_hi.y;
#pragma unroll
            for (int j = 0; j < kRows; j++) x ^= raw.row[j].x ^ raw.row[j].y ^ raw.row[j].z ^ raw.row[j].w;
            if (x == 0x12345679u) total_overflow[1] = x; // never true for real input, keeps the loads alive
        }
#pragma unroll
        for (int j = 0; j < (kParts >= 1 ? kRows / 2 : 0); j++)
        {
            uint32_t    t[8];
            const uint4 x = raw.row[j], y = raw.row[j + kRows / 2];
            rows_to_s2(x.x, y.x, t[0], t[1]);
            rows_to_s2(x.y, y.y, t[2], t[3]);
            rows_to_s2(x.z, y.z, t[4], t[5]);
            rows_to_s2(x.w, y.w, t[6], t[7]);
            uint4* dst = reinterpret_cast<uint4*>(&tile32[j * kRowSamples + 8 * lane]);
            dst[0]     = make_uint4(t[0], t[1], t[2], t[3]);
            dst[1]     = make_uint4(t[4], t[5], t[6], t[7]);
        }
        if constexpr (kParts >= 1)
        { // The continuation, 256 dwords, four per lane: the low halves repeat the start of the upper half, which the first pass above has
          // just written into the high halves of dwords 0 .. 255 -- read back from there (one 16-byte read per lane) --, the high halves are
          // the halo, squared here two samples of the same row per register.  16 vector instructions; as a fifth pass of the loop above
          // (row 4 once more beside row 8, half the lanes idle) it cost 40, and fetching row 4's samples a second time instead of
          // reading their squares back traded those for memory traffic the kernel cannot afford either.
            const uint32_t h01 = iq2_to_s2(raw.cont_hi.x), h23 = iq2_to_s2(raw.cont_hi.y);
            wave_lds_fence();
            const uint4 lo = *reinterpret_cast<const uint4*>(&tile32[4 * lane]);
            *reinterpret_cast<uint4*>(&tile32[kHalfChunk + 4 * lane]) =
                make_uint4(__builtin_amdgcn_perm(h01, lo.x, 0x05040302u), __builtin_amdgcn_perm(h01, lo.y, 0x07060302u),
                           __builtin_amdgcn_perm(h23, lo.z, 0x05040302u), __builtin_amdgcn_perm(h23, lo.w, 0x07060302u));
        }

        // ---------------- prefetch: the next chunk's loads fly while this chunk is processed
        const ChunkGeom cur = g;
        const uint32_t  me  = chunk;
        // control traffic first -- the previous chunk's directory entry and sums, the ticket for the work item after `next` --, the loads
        // behind it: what the wait at the top of the next trip covers was all issued a chunk's time before.  The ticket's value is read at
        // the end of this trip, half a chunk's time from here.  (Until round 4 a wave held a third work item in reserve, taken a whole trip
        // before its prefetch: a wave that drew one of its counter's last tickets then still had three chunks to go while its neighbours ran
        // dry, and the waves of a launch ended 30 us apart, 18 us idle on average; profiles/r04_sweep.txt.)
        publish(a, pend, lane);
        uint32_t ticket = 0;
        if (next != kNoChunk) ticket = grab_issue(a, wr, lane);
        if (next != kNoChunk)
        {
            g = chunk_geom_of(a, next, kFrameSpan);
            load_window<kHalo, false, true>(g, lane, raw);
            // (Touching the chunk after that one into the caches -- one dword per 64 bytes, two chunks ahead, so that the memory system has
            // requests while the wave computes -- was tried: the kernel went from 0.212 to 0.32 ms; profiles/r03_sweep.txt.  So was issuing
            // these loads row by row inside the image build, each into the register the build has just consumed -- a quarter of a chunk's
            // time earlier, for no register: 0.2155 -> 0.2260 ms; nine loads back to back are one 8.5 KB burst to the memory, four pairs
            // 0.3 us apart are not; profiles/r04_sweep.txt.)
        }
        wave_lds_fence();

        // ---------------- stage 1 on the interleaved image: 512 dwords per pass, a lane takes dwords q = 8 lane .. 8 lane + 7 of
        // them, i.e. eight positions q in the low halves and the eight positions q + 2048 in the high halves.  With
        // s_a = sample q + a of either position the ten comparisons (ADSB1090.cpp:782-783) are
        //     s0 > max(s1, s3, s4, s5, s6)    s2 > max(s1, s3)    s7 > s8    s9 > max(s6, s8)
        // and max(s_a, s_a+2) is one shared array (used at a = 1, 4 and 6).
        uint32_t surv32[2] = {0u, 0u};
#pragma unroll
        for (int b = 0; b < (kParts >= 2 ? kHalfChunk / 512 : 0); b++)
        {
            uint32_t        T[17];
            const uint32_t* p  = &tile32[b * 512 + 8 * lane];
            const uint4     q0 = *reinterpret_cast<const uint4*>(p), q1 = *reinterpret_cast<const uint4*>(p + 4);
            const uint4     q2 = *reinterpret_cast<const uint4*>(p + 8), q3 = *reinterpret_cast<const uint4*>(p + 12);
            T[0] = q0.x; T[1] = q0.y; T[2] = q0.z; T[3] = q0.w; T[4] = q1.x; T[5] = q1.y; T[6] = q1.z; T[7] = q1.w;
            T[8] = q2.x; T[9] = q2.y; T[10] = q2.z; T[11] = q2.w; T[12] = q3.x; T[13] = q3.y; T[14] = q3.z; T[15] = q3.w;
            // T[16] as the first dword of a fifth 16-byte read: a 4-byte read at this lane stride (32 bytes) hits eight banks only and takes
            // longer than the 16-byte one (tools/ldsbench.hip: 18 against 132 bytes per clock); written out, or the compiler narrows it
            // again.  The second statement is the wait: it redefines q4 as far as the compiler is concerned, so no use can be scheduled
            // in front of it (two statements so that the four reads above are not held up by it).
            {
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                u32x4_t q4;
                asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(q4) : "v"(tile_addr + 4u * (uint32_t)(b * 512 + 8 * lane)) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q4) : : "memory");
                T[16] = q4.x;
            }
            uint32_t M2[14]; // (max(s_a, s_a+2)) of both positions
#pragma unroll
            for (int a = 1; a <= 13; a++) M2[a] = pk_max(T[a], T[a + 2]);
            // Sign bit of each half of d1&d2&d3&d4 set <=> that position passes all ten comparisons.  Movemask by dot product:
            // with the flags isolated at bits 15 and 31, dot2(flags, (2^2k, 2^(2k+1))) adds 2^(15+2k) and 2^(16+2k), so the eight
            // dwords accumulate into bits 15..30 of one register, one VALU op per dword, in the order of the image's halves.
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                const uint32_t l0 = pk_max(pk_max(M2[k + 4], T[k + 5]), M2[k + 1]); // max(s1, s3, s4, s5, s6)
                const uint32_t d1 = pk_sub(l0, T[k]);                               // < 0 : all of them < s0
                const uint32_t d2 = pk_sub(M2[k + 1], T[k + 2]);                    // max(s1, s3) < s2
                const uint32_t d3 = pk_sub(T[k + 8], T[k + 7]);                     // s8 < s7
                const uint32_t d4 = pk_sub(M2[k + 6], T[k + 9]);                    // max(s6, s8) < s9
                const uint32_t ok = (d1 & d2 & d3 & d4) & 0x80008000u;
                const u16x2    wt = {(unsigned short)(1u << (2 * k)), (unsigned short)(2u << (2 * k))};
                acc               = __builtin_amdgcn_udot2(as_pk(ok), wt, acc, false);
            }
            // bit 2 k + h of the 16: position 2048 h + 512 b + 8 lane + k, i.e. the half at index 1024 b + 16 lane + (2 k + h) of the image
            // (acc holds nothing below bit 15 and nothing at bit 31: every term is a multiple of 2^15 and there are at most 2^16 - 1 of them)
            if (b & 1) surv32[b >> 1] |= acc << 1;
            else surv32[b >> 1] = acc >> 15;
        }
        // bit n of surv: the half at index 1024 (n >> 4) + 16 lane + (n & 15) of the image (tile_index of its position)
        uint64_t surv = (uint64_t)surv32[0] | ((uint64_t)surv32[1] << 32);
        // The rest of the chunk is short dependent chains (scalar work, LDS round trips, a few vector operations at a time); the other
        // waves of the SIMD are mostly in the vector-dense image and stage-1 phases.  With raised priority these chains issue as soon as
        // they are ready instead of queueing behind that work, the wave is back in a dense phase sooner, and the vector unit idles less:
        // 0.2439 -> 0.2333 ms per GiB (in-process A/B; priority 1, 2 and 3 measure the same).  Purely a scheduling hint.
        __builtin_amdgcn_s_setprio(1);
        if (cur.npos < (uint32_t)kChunk)
        { // last chunk of a buffer: positions at or beyond N-240 do not exist (ADSB1090.cpp:772)
#pragma unroll
            for (int b = 0; b < 4; b++)
            {
                const int n0 = (int)cur.npos - (512 * b + 8 * lane), n1h = n0 - kHalfChunk; // valid positions of the lane's low / high group
                const uint32_t k0 = n0 >= 8 ? 0x5555u : (n0 <= 0 ? 0u : (0x5555u & ((1u << (2 * n0)) - 1u)));
                const uint32_t k1 = n1h >= 8 ? 0xAAAAu : (n1h <= 0 ? 0u : (0xAAAAu & ((1u << (2 * n1h)) - 1u)));
                surv &= ~(0xFFFFull << (16 * b)) | ((uint64_t)(k0 | k1) << (16 * b));
            }
        }

        // ---------------- survivors -> queue -> stage 2 -> demod, at most kQueueCap survivors per pass
        const uint32_t mine = (uint32_t)__builtin_popcountll(surv);
        const uint32_t incl = wave_incl_scan_add(mine);
        const uint32_t n1   = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        Emit           e = begin_chunk(a, me);
        if (kParts < 3) e.count = (n1 == 0xFFFFFFFFu) ? 1u : 0u; // part builds: keep what was computed alive, emit nothing
        for (uint32_t base = 0; kParts >= 3 && base < n1; base += (uint32_t)kQueueCap)
        {
            if (n1 <= (uint32_t)kQueueCap) queue_survivors<false>(surv, incl - mine, lane, queue_addr, 0u);
            else queue_survivors<true>(surv, incl - mine, lane, queue_addr, base);
            wave_lds_fence();
            const uint32_t nq = (n1 - base < (uint32_t)kQueueCap) ? (n1 - base) : (uint32_t)kQueueCap;

            // stage 2 (:794-811), dense over lanes; the lanes that pass hand their candidate to the whole wave one after the other
            for (uint32_t qb = 0; qb < nq; qb += 64)
            {
                const uint32_t idx = qb + (uint32_t)lane;
                bool           ok  = false, unsure = false;
                uint32_t       ti  = 0;
                if (idx < nq)
                {
                    // high = (m0+m2+m7+m9)/6 needs four exact magnitudes; "m_x < high" for the six quiet samples is then
                    // one test on the largest of their s values: m(s) <= high-1  <=>  129600*s <= high^2 - high.
                    ti                  = queue[idx];
                    const uint16_t* w   = &tile[ti]; // sample a of this position: w[2 a]
                    uint32_t       sq   = w[2 * 4];
                    sq = (w[2 * 5] > sq) ? w[2 * 5] : sq;
                    sq = (w[2 * 11] > sq) ? w[2 * 11] : sq;
                    sq = (w[2 * 12] > sq) ? w[2 * 12] : sq;
                    sq = (w[2 * 13] > sq) ? w[2 * 13] : sq;
                    sq = (w[2 * 14] > sq) ? w[2 * 14] : sq;
                    const uint32_t s0 = w[0], s2 = w[2 * 2], s7 = w[2 * 7], s9 = w[2 * 9];
                    // First on estimates: every magnitude estimate is within kEstErr of the reference's integer, so the sum of
                    // four is within 4 kEstErr and high = sum / 6 (truncating) lies in [(sum - 5) / 6, sum / 6].
                    //   surely quiet  : e(max) + kEstErr <= (sum_est - 4 kEstErr - 5) / 6 - 1
                    //   surely not    : e(max) - kEstErr >= (sum_est + 4 kEstErr) / 6
                    const float sum_est = 360.0f * (__builtin_amdgcn_sqrtf((float)s0) + __builtin_amdgcn_sqrtf((float)s2) +
                                                    __builtin_amdgcn_sqrtf((float)s7) + __builtin_amdgcn_sqrtf((float)s9));
                    const float max_est = mag_estimate(sq);
                    const bool  quiet   = max_est + kEstErr <= (sum_est - 4.0f * kEstErr - 5.0f) * (1.0f / 6.0f) - 1.0f - 0.01f;
                    const bool  loud    = max_est - kEstErr >= (sum_est + 4.0f * kEstErr) * (1.0f / 6.0f) + 0.01f;
                    ok                  = quiet;
                    unsure              = !(quiet || loud);
                }
                if (ballot(unsure))
                { // some lane is too close to call: the exact test for the whole pass
                    ok = false;
                    if (idx < nq)
                    {
                        const uint16_t* w   = &tile[ti];
                        const uint32_t high = (uint32_t)(mag_of_s(w[0]) + mag_of_s(w[2 * 2]) + mag_of_s(w[2 * 7]) + mag_of_s(w[2 * 9])) / 6u;
                        uint32_t       sq   = w[2 * 4];
                        sq = (w[2 * 5] > sq) ? w[2 * 5] : sq;
                        sq = (w[2 * 11] > sq) ? w[2 * 11] : sq;
                        sq = (w[2 * 12] > sq) ? w[2 * 12] : sq;
                        sq = (w[2 * 13] > sq) ? w[2 * 13] : sq;
                        sq = (w[2 * 14] > sq) ? w[2 * 14] : sq;
                        const uint32_t se = sq + ((sq + 1u) >> 15);
                        ok = high != 0 && (uint32_t)__umul24(se, 129600u) <= high * high - high;
                    }
                }
                uint64_t mk = ballot(ok);
                if (kParts < 4) e.count += (mk == 0x123456789ull) ? 1u : 0u;
                // demodulate the candidates, one at a time, whole wave each (the order inside a chunk is the ordering pass's business)
                while (kParts >= 4 && mk)
                {
                    const int      src = __builtin_ctzll(mk);
                    mk &= mk - 1ull;
                    const uint32_t a0  = (uint32_t)__builtin_amdgcn_readlane((int)ti, src);
                    const uint32_t pos = (a0 >> 1) + (a0 & 1u) * (uint32_t)kHalfChunk;
                    Win            w;
                    w.a0  = a0;
                    // the sample in front of position 0 is the one in front of the chunk; the one in front of position 2048 is the last low half
                    w.am1 = a0 == 0 ? (uint32_t)kFrontSlot16 : (a0 == 1u ? (uint32_t)(2 * (kHalfChunk - 1)) : a0 - 2u);
                    const uint8_t* front = cur.buf + 2ull * (cur.g0 - 1u); // only dereferenced for position 0 of a chunk that is not the buffer's first
                    const int      todo  = demod_strong_frame(tile, tile_addr, lane, lt, fc, e, w, cur.g0 + pos, front);
                    if (todo) demod_candidate(tile, lane, lt, e, w, cur.g0 + pos, todo == 2, front);
                }
            }
            wave_lds_fence();
        }
        pend = finish_chunk(me, e);

        if (next == kNoChunk) break;
        chunk = next;
        next  = take_next(a, wr, ticket, lane);
    }
    publish(a, pend, lane);
    flush_records();
    stamp(a.stamps, 1);
}

// ---------------------------------------------------------------------------------------------
// The ordering pass on its own (gather1090.hip.h has the routine and the reasons): for a scan that no later scan kernel on its stream orders --
// the last step of a loop, a live buffer, a serial caller.  One wave per workgroup, blocks drawn from the slot's counter.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gather_kernel(GatherArgs ga)
{
    __shared__ __attribute__((aligned(16))) GatherLds lds;
    stamp(ga.stamps, 2);
    gather_units(ga, blockIdx.x, gridDim.x, 4u, lds, threadIdx.x); // (a quarter block per wave: nothing else runs beside this kernel)
    stamp(ga.stamps, 3);
}

// parity helper: the same decoder over an arbitrary record array
__global__ __launch_bounds__(256) void decode1090_kernel(const adsb_amd_record_t* __restrict__ rec, adsb_amd_decoded_t* __restrict__ out, size_t n)
{
    __shared__ uint8_t ais[64];
    ais_table_init(ais, threadIdx.x);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t B[3];
    for (int k = 0; k < 3; k++)
        B[k] = ((uint32_t)rec[i].msg[4 * k] << 24) | ((uint32_t)rec[i].msg[4 * k + 1] << 16) | ((uint32_t)rec[i].msg[4 * k + 2] << 8) | (uint32_t)rec[i].msg[4 * k + 3];
    const FieldsDev d = decode_fields_dev(B[0], B[1], B[2], rec[i].df, ais); // what the ordering pass runs
    *reinterpret_cast<uint4*>(out + i) = make_uint4(d.head, d.altitude, d.a, d.b);
}

// ---------------------------------------------------------------------------------------------
// parity helpers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void magnitude1090_kernel(const uint8_t* __restrict__ iq, uint16_t* __restrict__ mag, size_t n)
{
    // same device functions as the scan kernel: packed s for two samples, then the exact magnitude of each
    const size_t tid    = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p = tid; p < n / 2; p += stride)
    {
        const uint8_t* b  = iq + 4 * p;
        const uint32_t x  = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        const uint32_t s2 = iq2_to_s2(x);
        mag[2 * p]        = (uint16_t)mag_of_s(s2 & 0xFFFFu);
        mag[2 * p + 1]    = (uint16_t)mag_of_s(s2 >> 16);
    }
    if (tid == 0 && (n & 1u)) mag[n - 1] = (uint16_t)mag_of_s(iq1_to_s(iq[2 * (n - 1)], iq[2 * (n - 1) + 1]));
}

__global__ __launch_bounds__(256) void phase978_kernel(const uint8_t* __restrict__ iq, uint16_t* __restrict__ phi, size_t n,
                                                       const uint16_t* __restrict__ lut)
{
    size_t i      = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) phi[i] = lut[(uint32_t)iq[2 * i] | ((uint32_t)iq[2 * i + 1] << 8)];
}

} // namespace

hipError_t launch_scan1090(const ScanArgs& a, uint32_t* total_and_overflow, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const GatherArgs* attached)
{
    if (a.total_chunks == 0) return hipMemsetAsync(total_and_overflow, 0, kStateWords * sizeof(uint32_t), stream);
    // persistent single-wave workgroups: enough to fill every CU at the LDS-limited occupancy (16 per CU)
    const uint32_t   grid = scan_grid(a);
    const GatherArgs ga   = attached ? *attached : GatherArgs{};
    if (start || stop) hipExtLaunchKernelGGL(scan1090_kernel, dim3(grid), dim3(64), 0, stream, start, stop, 0, a, total_and_overflow, ga); // (either may be NULL)
    else hipLaunchKernelGGL(scan1090_kernel, dim3(grid), dim3(64), 0, stream, a, total_and_overflow, ga);
    return hipGetLastError();
}

hipError_t launch_decode1090(const adsb_amd_record_t* rec, adsb_amd_decoded_t* out, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(decode1090_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, rec, out, n);
    return hipGetLastError();
}

hipError_t launch_gather1090(const GatherArgs& ga, hipStream_t stream, hipEvent_t done)
{
    if (ga.nblocks == 0) return hipSuccess;
    const uint32_t units = ga.nblocks * 4u /* kGatherSplit */, grid = units < 8192u ? units : 8192u;
    if (done) hipExtLaunchKernelGGL(gather_kernel, dim3(grid), dim3(64), 0, stream, nullptr, done, 0, ga);
    else hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(64), 0, stream, ga);
    return hipGetLastError();
}

hipError_t launch_magnitude1090(const uint8_t* iq, uint16_t* mag, size_t nsamples, hipStream_t stream)
{
    if (nsamples == 0) return hipSuccess;
    size_t blocks = (nsamples / 2 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(magnitude1090_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, iq, mag, nsamples);
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
