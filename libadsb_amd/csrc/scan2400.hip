// scan2400.hip -- gfx950 kernel of the 2.4 MS/s Mode S scan mode (ADSB_AMD_MODE_2400).
//
// Not a counterpart of anything libadsb compiles: the reference demodulates 2 samples per microsecond only (ADSB1090.cpp:148,
// 757-758).  BASELINE.json quotes its metric on "2.4 MS/s u8 IQ", so the library has this second mode; its
// definition is oracle/oracle2400.c (read that header for the signal geometry and every rule), and this kernel has to produce
// exactly the records that file produces -- parity unpinned, GPU == specification, generator -> decoder round trips.
//
// Shape, the 2 MS/s kernel's: persistent single-wave workgroups walk XCD-local chunk ranges off work counters, the next chunk's
// window is prefetched into registers while the current one is processed:
//   * a wave owns 4096 preamble positions of one buffer, loads the 4416 samples they can touch with 16-byte coalesced loads and
//     parks s = (I-127)^2 + (Q-127)^2 as a LINEAR image of halves in LDS (sample p of the chunk at half p + 8);
//   * the gate (oracle2400_gate) runs in two steps since round 5.  Dense, on the MATRIX pipe: a necessary condition that is linear in s,
//         (s0+s1) + (s2+s3) + (s8+s9) > 2 (s-1 + s5+s6+s7 + s14+s15+s16+s17)
//     (3 min(A,B,C,D) > 2 Q implies 3 (A+B+C)/3 > 2 Q), evaluated for 1024 positions at a time as a banded (Toeplitz) matrix product with
//     v_mfma_f32_32x32x16_f16: the image itself is the B operand -- 32 columns, column c = the 64 samples from 32 c - 8 on, converted to f16 on
//     the way in --, the A operand holds the filter's weights for the 32 positions of a column (a 4 KB device table, fetched per chunk).  f16 x f16
//     products are exact in f32; what is not exact (the conversion of a 15-bit s to eleven bits of mantissa, the f32 summation) is covered the safe
//     way round: weights 1 + 2^-10 and -2 (1 - 2^-11), accumulators started at +4, so that the sign says "fail" only where the exact sum fails.
//     The vector unit converts and collects 32 sign bits per register: 3.1 instructions per 64 positions where the packed 16-bit gate took 8.3.
//     (The exact integer form -- v_mfma_i32_32x32x32_i8 over the low and the high bytes of the halves, two chains -- was built first and is not
//     what runs: no faster than the packed gate, profiles/r05_mode2400_variants.txt.)  The survivors of that condition (about forty per chunk) go
//     through the exact gate sparsely, a lane each;
//   * the preamble correlation of the five sub-sample phases for sixteen gate survivors at a time, exactly, as an i8 matrix product
//     (v_mfma_i32_32x32x32_i8 over the bytes of the survivors' exact magnitudes: kScoreTable) -- most gate survivors of noise end there;
//   * a candidate is demodulated by the whole wave: per phase tried lane b slices bit b and bit 64 + b from four samples with the overlap weights of its own
//     sub-sample position -- on float estimates of the magnitudes, exactly only where a decision lies inside the estimates' error
//     margin --, ballots give the message, parity is the DPP XOR reduction of per-lane table entries, the one-bit repair a
//     ballot over per-lane syndromes (shared with the 2 MS/s kernel).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <stdint.h>

#include "scan1090.h"
#include "gather1090.hip.h"
#include "scan_common.hip.h"

namespace adsb_amd
{
namespace
{

constexpr int kSpan24    = 292;                       // samples after j a candidate may read (oracle2400.c)
constexpr int kHalo24    = 320;                       // halo samples of the image (>= kSpan24, a multiple of 8)
constexpr int kImgPad    = 8;                         // halves in front of the chunk's first sample: sample p lives at half p + kImgPad, so that a lane's
                                                      // eight samples land on a 16-byte boundary AND a column of the matrix product starts on one
                                                      // (only the last of the eight, the sample before the chunk, is ever read with a non-zero weight)
constexpr int kImgHalves = kImgPad + kChunk + kHalo24; // 4424 halves = 8848 bytes
constexpr int kQueue24   = 64;                        // queue entries per pass (a lane holds at most 64)
constexpr int kGateBlock = 1024;                      // positions per matrix product: 32 columns of 32 positions
static_assert(kChunk % kGateBlock == 0 && kImgHalves % 8 == 0, "whole blocks, 16-byte rows");
// the last column of the last block reads 64 samples from sample 4096 - 32 - 8 on: inside the halo
static_assert(kChunk - 32 - kImgPad + 64 <= kChunk + kHalo24, "the matrix product's reads stay inside the image");

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float    f16_t __attribute__((ext_vector_type(16)));

// Weight of tap a (sample j + a) in the necessary condition A + B + C - 2 Q > 0 of position j
constexpr int prefilter_weight(int a)
{
    return (a == 0 || a == 1 || a == 2 || a == 3 || a == 8 || a == 9) ? 1 : (a == -1 || a == 5 || a == 6 || a == 7 || (a >= 14 && a <= 17)) ? -2 : 0;
}
// The product runs on v_mfma_f32_32x32x16_f16: the halves of the image are converted to f16 on the way in (round to nearest: off by at most
// 2^-11 of the value), the products of two f16 numbers are exact in f32, the sum is an f32 sum.  The condition stays NECESSARY because the weights
// lean the right way -- a pulse sample counts 1 + 2^-10 (s <= fl(s) (1 + 2^-10)), a quiet one 2 (1 - 2^-11) (s >= fl(s) (1 - 2^-11)) -- and the
// accumulators start from +4, more than sixty-four f32 additions of numbers below 2^20 can lose.  (An integer product -- i8, low and high bytes of
// the halves as two chains, exact -- was built first: twice the matrix instructions, twice the LDS reads, a shift, an XOR and a combine per
// register where this form has the conversions; measured no faster than the packed gate it replaced, profiles/r05_mode2400_variants.txt.)
constexpr uint32_t kPulseWeightBits = 0x3C01u; // f16 1 + 2^-10
constexpr uint32_t kQuietWeightBits = 0xBFFFu; // f16 -2 (1 - 2^-11) = -1.9990234375: the step from -2 towards zero

// overlap, in fifths of a sample, of the half-microsecond slot k of a frame that starts phi fifths into its first sample with
// sample t of the window (oracle2400.c: overlap5 / slot_energy)
constexpr int slot_overlap(int phi, int k, int t)
{
    const int lo = phi + 6 * k, hi = lo + 6, a = lo > 5 * t ? lo : 5 * t, b = hi < 5 * t + 5 ? hi : 5 * t + 5;
    return b > a ? b - a : 0;
}
// per phase and sample: weight of the sample in P(phi) (pulse slots 0, 2, 7, 9 minus quiet slots 1, 3, 4, 5, 6, 8).
// 13 samples cover every slot up to 9 for every phase.
struct PreambleWeights
{
    int8_t p[5][13];
};
constexpr PreambleWeights make_preamble_weights()
{
    PreambleWeights w{};
    for (int phi = 0; phi < 5; phi++)
        for (int t = 0; t < 13; t++)
        {
            int pulse = 0, quiet = 0;
            for (int k : {0, 2, 7, 9}) pulse += slot_overlap(phi, k, t);
            for (int k : {1, 3, 4, 5, 6, 8}) quiet += slot_overlap(phi, k, t);
            w.p[phi][t] = (int8_t)(pulse - quiet);
        }
    return w;
}
constexpr PreambleWeights kPreambleHost = make_preamble_weights();

// The preamble scores as a matrix product (round 5).  Seven weighted sums of a gate survivor's 13 window magnitudes are wanted -- P(phi) for the
// five phases, m0 + .. + m11 and m12 - m0 -- exactly, for every survivor.  The magnitudes (16 bits) go to LDS as bytes, survivor by survivor,
// and v_mfma_i32_32x32x32_i8 forms all sums of sixteen survivors at once: column = survivor, K = the 32 bytes of its 16 halves (13 magnitudes,
// one constant, two unused), rows = (sum, byte plane).  The matrix cores multiply SIGNED bytes, so every byte is stored XOR 0x80 (= minus 128);
// half 13 holds the constant bytes 0x80 0x80 (minus 128 each) with weight minus (sum of the row's weights), which takes the bias out again:
// row (f, low) is exactly sum w_f[t] low(m_t), row (f, high) exactly sum w_f[t] high(m_t), and the sum is 256 high + low.
// A lane of the result holds rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): the low-byte row of sum f is given register f, its high-byte row
// register 8 + f, both in the lower lane half, so that lane n (n < 16) ends up with everything about survivor n in its own registers.
// Replaces four survivors per trip with one magnitude per lane, seven DPP row sums and a best-of-five per row of sixteen lanes.
struct ScoreTable
{
    uint32_t w[64][4]; // [lane = row | half << 5][dword of its 16 K bytes]
};
constexpr int score_weight(int f, int t)
{ // weight of half t of a survivor's row in sum f (t = 13: the constant half)
    int w = 0, sum = 0;
    for (int u = 0; u < 13; u++)
    {
        const int x = f < 5 ? kPreambleHost.p[f][u] : f == 5 ? (u < 12 ? 1 : 0) : (u == 12 ? 1 : u == 0 ? -1 : 0);
        sum += x;
        if (u == t) w = x;
    }
    return t < 13 ? w : t == 13 ? -sum : 0;
}
constexpr ScoreTable make_score_table()
{
    ScoreTable t{};
    for (int lane = 0; lane < 64; lane++)
    {
        const int r = lane & 31, h = lane >> 5;
        if (r & 4) continue;                       // rows of the upper lane half's registers: unused
        const int reg = (r & 3) + 4 * (r >> 3);     // the register this row comes out in (lower lane half)
        const int f = reg & 7, plane = reg >> 3;    // sum, byte plane (0 low, 1 high)
        if (f >= 7) continue;
        for (int b = 0; b < 16; b++)
        {
            const int k = 16 * h + b, half = k >> 1;
            if ((k & 1) != plane) continue;
            t.w[lane][b >> 2] |= ((uint32_t)score_weight(f, half) & 0xFFu) << (8 * (b & 3));
        }
    }
    return t;
}
__device__ const ScoreTable kScoreTable = make_score_table();

// Correlation of bit b at phase phi (oracle2400.c: c_b): the four samples of its window, weights of its own sub-sample position p,
// 5 i0 + p = phi + 96 + 12 b: {5-p, 2p-3, -min(2+p,5), -(p==4)}.  Two forms.  The float one works on estimates of the magnitudes (each within
// kEstErr = 1.6 of the exact one, scan_common.hip.h), so it is within 12 * 1.6 = 19.2 of the exact correlation whatever p is; the integer one
// computes the four exact magnitudes.  A candidate is sliced on estimates, and exactly only when some bit's sign, or its place relative to the
// weak-bit line 2 |c| < A, lies inside that margin -- a frame well above the noise never gets there, and the 192-296 exact magnitudes per
// candidate of rounds 1-2 (two thirds of a candidate's instructions) are not computed at all.
constexpr float kCorrErr = 19.5f;
struct BitWindow
{
    uint32_t s[4];
    int      p;
};
__device__ __forceinline__ BitWindow bit_window(const uint16_t* img16, uint32_t a0, int phi, int b)
{
    const int T  = phi + 96 + 12 * b;
    const int i0 = T / 5;
    BitWindow w;
    w.p = T - 5 * i0;
#pragma unroll
    for (int t = 0; t < 4; t++) w.s[t] = img16[a0 + (uint32_t)(i0 + t)];
    return w;
}
__device__ __forceinline__ float corr_estimate(const BitWindow& w)
{
    const float p  = (float)w.p; // the weights in float arithmetic (exact: small whole numbers)
    const float w0 = 5.0f - p, w1 = 2.0f * p - 3.0f, w2 = __builtin_fminf(2.0f + p, 5.0f), w3 = w.p == 4 ? 1.0f : 0.0f;
    return w0 * mag_estimate(w.s[0]) + w1 * mag_estimate(w.s[1]) - w2 * mag_estimate(w.s[2]) - w3 * mag_estimate(w.s[3]);
}
__device__ __forceinline__ int corr_exact(const BitWindow& w)
{
    const int p = w.p, w2 = (2 + p < 5) ? 2 + p : 5;
    return (5 - p) * mag_of_s(w.s[0]) + (2 * p - 3) * mag_of_s(w.s[1]) - w2 * mag_of_s(w.s[2]) - (p == 4 ? mag_of_s(w.s[3]) : 0);
}
// bits (c > 0) and weak bits (2 |c| < amp) of 64 bit positions, lane = position: on the estimates, with the lanes whose answer lies inside the
// margin in `unsure`; and exactly
__device__ __forceinline__ void bits_estimated(const BitWindow& w, int amp, uint64_t& val, uint64_t& weak, uint64_t& unsure)
{
    const float c = corr_estimate(w), mag = __builtin_fabsf(c), half = 0.5f * (float)amp;
    val           = ballot(c > 0.0f);
    weak          = ballot(mag < half);
    unsure        = ballot(!(mag > kCorrErr && (mag + kCorrErr < half || mag - kCorrErr >= half)));
}
__device__ __forceinline__ void bits_exact(const BitWindow& w, int amp, uint64_t& val, uint64_t& weak)
{
    const int c = corr_exact(w);
    val         = ballot(c > 0);
    weak        = ballot(2 * abs(c) < amp);
}

// One slice of the window at phase phi (wave-uniform): records what oracle2400.c's slice_phase accepts (the rejections in a
// cheaper order: the DF before the second half is even looked at).  Returns true when a record was emitted.
__device__ __forceinline__ bool slice_and_emit(const uint16_t* img16, uint32_t a0, int lane, const LaneTables& lt, Emit& e, uint32_t j, int phi, int amp)
{
    const BitWindow wA = bit_window(img16, a0, phi, lane);
    uint64_t        valA, wkA, unsureA;
    bits_estimated(wA, amp, valA, wkA, unsureA);
    // the first 56 bits decide everything about a short frame and the DF of any; bits 56..63 count only for a long one
    bool exactA = (unsureA & kMask56) != 0;
    if (exactA) bits_exact(wA, amp, valA, wkA);
    const uint32_t df      = (uint32_t)(__builtin_bitreverse64(valA) >> 59);
    const bool     is17    = (df == 11 || df == 17);
    if (!is17 && !df_is_ap(df)) return false;
    const bool     is_long = df_is_long(df);
    const uint32_t nbits   = is_long ? 112u : 56u;
    const bool     has_b   = lane < 48;
    uint64_t       ba = valA & kMask56, bb = 0ull;
    int            weak = __builtin_popcountll(wkA & kMask56);
    if (is_long)
    {
        if (!exactA && (unsureA >> 56) != 0) bits_exact(wA, amp, valA, wkA);
        const BitWindow wB = bit_window(img16, a0, phi, has_b ? 64 + lane : 64);
        uint64_t        wkB, unsureB;
        bits_estimated(wB, amp, bb, wkB, unsureB);
        if ((unsureB & kMask48) != 0) bits_exact(wB, amp, bb, wkB);
        ba = valA;
        bb &= kMask48;
        weak = __builtin_popcountll(wkA) + __builtin_popcountll(wkB & kMask48);
    }
    if (weak > (int)nbits / 8) return false;
    uint32_t contrib, stored;
    if (is_long)
    {
        contrib = (((ba >> lane) & 1ull) ? lt.crc_a : 0u) ^ ((has_b && ((bb >> lane) & 1ull)) ? lt.crc_b : 0u);
        stored  = (uint32_t)(__builtin_bitreverse64(bb) >> 16) & 0xFFFFFFu;
    }
    else
    {
        contrib = ((ba >> lane) & 1ull) ? lt.crc_s : 0u; // crc_s is 0 on lanes >= 56
        stored  = (uint32_t)(__builtin_bitreverse64(ba) >> 8) & 0xFFFFFFu;
    }
    const uint32_t syn = wave_xor(contrib) ^ stored;
    if (is17)
    {
        int errorbit = -1;
        if (syn != 0)
        {
            if (weak > 2) return false;
            // the syndromes of flipping a bit, worked out here, in the one frame in ten that needs a repair, from a lane number the compiler cannot see
            // through (LaneTables::syn_b / syn_s, scan_common.hip.h)
            int ol = lane;
            asm volatile("" : "+v"(ol));
            const uint32_t syn_b = lt.syn_b(ol), syn_s = lt.syn_s(ol);
            const uint64_t ma = is_long ? ballot(syn == lt.crc_a) : ballot(ol < 56 && syn == syn_s);
            const uint64_t mb = is_long ? ballot(ol < 48 && syn == syn_b) : 0ull;
            if (ma) errorbit = __builtin_ctzll(ma);
            else if (mb) errorbit = 64 + __builtin_ctzll(mb);
            else return false;
        }
        emit_raw(e, lane, j, ba, bb, df, nbits, errorbit, 0u, 0u, (uint32_t)phi);
        return true;
    }
    if (weak > 4) return false;
    emit_raw(e, lane, j, ba, bb, df, nbits, -1, ADSB_AMD_F_NEEDS_ICAO, syn, (uint32_t)phi);
    return true;
}

// Two samples' I, Q bytes (I0 Q0 I1 Q1) -> (s0 | s1 << 16): byte ^ 0x7F is 127 - byte as a signed 8-bit number (scan_common.hip.h, rows_to_s2);
// v_perm_b32 sign-extends the odd bytes of its second operand (selectors 8 and 9), the even ones after a shift by one byte
__device__ __forceinline__ uint32_t iq2_to_s2_linear(uint32_t x)
{
    const uint32_t z = x ^ 0x7F7F7F7Fu, u = z << 8;
    return pair_to_s2(__builtin_amdgcn_perm(0u, u, 0x09030801u), __builtin_amdgcn_perm(0u, z, 0x09030801u));
}

// The A operand of the prefilter's matrix product: lane l = (row r = l & 31, half h = l >> 5) holds, for each of the four 16-sample steps kb
// of a column's 64 samples, the weights of samples 16 kb + 8 h .. + 7 of the column (sample m of column c is sample 32 c - 8 + m of the chunk)
// for the position its row stands for.
// Row r of the product comes out in register i = (r & 3) + 4 (r >> 3) of lane half (r >> 2) & 1 (the instruction's result map); it is given
// position v = 16 ((r >> 2) & 1) + 4 (r >> 3) + (r & 3) of the column, so that lane (c, h) ends up with positions 32 c + 16 h + i, i = 0 .. 15,
// in register order: sixteen consecutive bits of the survivor map.
// The weights are a table in device memory, 64 bytes a lane, fetched at the top of every chunk and dead after the product: kept in registers
// across the chunk's sparse phases they were sixteen registers too many (a row of the prefetched window went to scratch and back).
struct GateTable
{
    uint32_t w[64][16]; // [lane][4 kb + j]: f16 pair of samples 16 kb + 8 h + 2 j, + 1
};
constexpr GateTable make_gate_table()
{
    GateTable t{};
    for (int lane = 0; lane < 64; lane++)
    {
        const int r = lane & 31, h = lane >> 5, v = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
        for (int kb = 0; kb < 4; kb++)
            for (int j = 0; j < 8; j++)
            {
                const int      a = -kImgPad + 16 * kb + 8 * h + j - v; // the sample's offset from the row's position
                const int      w = (a >= -1 && a <= 17) ? prefilter_weight(a) : 0;
                const uint32_t b = w > 0 ? kPulseWeightBits : w < 0 ? kQuietWeightBits : 0u;
                t.w[lane][4 * kb + (j >> 1)] |= b << (16 * (j & 1));
            }
    }
    return t;
}
__device__ const GateTable kGateTable = make_gate_table();

// (two conversions per register, the second into the upper half of the first one's result; left to itself the compiler converts into two
// registers and packs them: three instructions)
__device__ __forceinline__ uint32_t pair_to_f16(uint32_t x)
{
    uint32_t r;
    asm("v_cvt_f16_u16_e32 %0, %1\n\t"
        "v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1"
        : "=&v"(r)
        : "v"(x));
    return r;
}
__device__ __forceinline__ h8_t halves_to_f16(uint4 x)
{
    typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(h8_t, u4_t{pair_to_f16(x.x), pair_to_f16(x.y), pair_to_f16(x.z), pair_to_f16(x.w)});
}

__global__ __launch_bounds__(64, 4) void scan2400_kernel(ScanArgs a, uint32_t* __restrict__ total_overflow, GatherArgs ga)
{
    __shared__ __attribute__((aligned(16))) uint16_t img16[kImgHalves];
    __shared__ __attribute__((aligned(16))) uint16_t gatemap[kChunk / 16]; // bit p of the map: position p passes the prefilter (64 x 64 bits)
    __shared__ uint16_t                              gqueue[kQueue24];    // prefilter survivors of a pass (positions)
    __shared__ uint16_t                              queue[kQueue24];     // gate survivors of a pass
    __shared__ uint8_t                               wlist[kQueue24];
    __shared__ uint32_t                              score[kQueue24], pulse[kQueue24]; // per queue entry: best P << 3 | phase; pulse amplitude A

    uint16_t* const scorebuf = gatemap; // sixteen survivors x 16 halves: the map is dead once every lane has read its 64 bits (see the scores)
    static_assert(sizeof(gatemap) >= 16 * 16 * sizeof(uint16_t), "the score buffer fits where the survivor map was");

    const int lane = threadIdx.x;
    stamp(a.stamps, 0);
    const LaneTables lt = load_lane_tables(a.crc_tab, lane);
    typedef int sw_t __attribute__((ext_vector_type(4)));
    const sw_t score_w = *reinterpret_cast<const sw_t*>(kScoreTable.w[lane]); // the scores' weights (A operand), four registers for the whole launch
    const int tl = lane & 15, rw = lane >> 4; // scoring: row rw of 16 lanes takes a survivor, lane tl of the row its window sample tl
    if (blockIdx.x == 0) gather_state_zero(total_overflow, (uint32_t)lane); // this slot's ordering state (GatherArgs::state): its pass runs in a later kernel on the stream
    // the ordering pass of the other slot's scan, by a quarter of this kernel's waves, a block each, before their own chunks (gather1090.hip.h; the image area is free until then)
    static_assert(sizeof(img16) >= sizeof(GatherLds), "the ordering pass's scratch fits in the image area");
    gather_in_front(ga, a.ncu, *reinterpret_cast<GatherLds*>(img16), (uint32_t)lane);
    if (lane < kImgPad) img16[lane] = 0; // the halves in front of the image: zero weights, but they go through the multiplier

    // persistent waves, chunk order and work counters as in scan1090_kernel (scan_common.hip.h)
    WorkRange wr = work_range(a);
    if (wr.slot >= wr.end) return;
    uint32_t  chunk = wr.chunk_of(wr.slot);
    uint32_t  next  = wr.slot + wr.nslot < wr.end ? wr.chunk_of(wr.slot + wr.nslot) : kNoChunk;
    ChunkGeom g     = chunk_geom_of(a, chunk, kSpan24);
    RawWindow raw;
    load_window<kHalo24>(g, lane, raw);
    Pending  pend{};     // the previous chunk's directory entry and sums, not yet written (scan_common.hip.h)

    for (;;)
    {
    // the prefilter's weights: issued here (nothing else is in flight: the window these rows came from has landed), used behind the image build.
    // The empty statement makes the address a new value every trip: a loop-invariant load would be hoisted and its registers stay live.
    // (global loads, said explicitly: through a generic pointer they would be flat loads, which may return out of order, and the wait for
    // the window's rows below would become a wait for these as well)
    typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u4_t* gw_ptr_t;
    uint64_t waddr = reinterpret_cast<uint64_t>(&kGateTable.w[lane][0]);
    asm volatile("" : "+v"(waddr));
    const gw_ptr_t wsrc = reinterpret_cast<gw_ptr_t>(waddr);
    const u4_t     wq0 = wsrc[0], wq1 = wsrc[1], wq2 = wsrc[2], wq3 = wsrc[3];
    // ---------------- window -> s: a lane's sixteen bytes of a row are eight consecutive samples, eight halves of the image
    __builtin_amdgcn_s_setprio(0);
    wave_lds_fence(); // readers of the previous chunk are done
#pragma unroll
    for (int jr = 0; jr <= kRows; jr++)
    {
        const uint4 x = raw.row[jr];
        if (jr < kRows || lane < kHalo24 / 8)
            *reinterpret_cast<uint4*>(&img16[kImgPad + jr * kRowSamples + 8 * lane]) =
                make_uint4(iq2_to_s2_linear(x.x), iq2_to_s2_linear(x.y), iq2_to_s2_linear(x.z), iq2_to_s2_linear(x.w));
    }
    // the sample before the chunk (0 at the start of a buffer: oracle2400_gate)
    if (lane == 0) img16[kImgPad - 1] = (uint16_t)iq1_to_s(raw.front & 0xFFu, raw.front >> 8);

    // (the weights, fetched at the top of the trip, are claimed here: left to the product, the compiler's wait for them would sit behind the
    // prefetch, and where the paths of load_window meet it counts the loads of the shortest one -- a wait for most of the window)
    asm volatile("" ::"v"(wq0), "v"(wq1), "v"(wq2), "v"(wq3));
    // ---------------- prefetch: the next chunk's loads fly while this chunk is processed (the whole of it: issued behind the prefilter instead, the
    // window had not landed when the next trip wanted it -- the sparse phases that were left to cover the loads are too short)
    const uint32_t me = chunk, g0 = g.g0, npos = g.npos;
    // control traffic (the previous chunk's directory entry and sums, the ticket for the work item after `next`) in front of the loads
    publish(a, pend, lane);
    uint32_t ticket = 0;
    if (next != kNoChunk) ticket = grab_issue(a, wr, lane);
    if (next != kNoChunk)
    {
        g = chunk_geom_of(a, next, kSpan24);
        int pl = lane; // (a new value every trip: the loads' lane offsets are recomputed -- three instructions -- rather than kept in scratch)
        asm volatile("" : "+v"(pl));
        load_window<kHalo24>(g, pl, raw);
    }
    wave_lds_fence(); // the image is complete

    // ---------------- prefilter on the matrix pipe (see the head of the file): 1024 positions per trip.  Lane (c, h) reads samples
    // 16 kb + 8 h .. + 7 of column c (one ds_read_b128 per step; at a lane stride of 64 bytes they are four-way bank conflicted, which the
    // LDS has room for) and converts them.
    {
        const uint4* col   = reinterpret_cast<const uint4*>(img16) + 4 * (lane & 31) + (lane >> 5);
        const h8_t   gw[4] = {__builtin_bit_cast(h8_t, wq0), __builtin_bit_cast(h8_t, wq1), __builtin_bit_cast(h8_t, wq2), __builtin_bit_cast(h8_t, wq3)};
#pragma unroll 1
        for (int blk = 0; blk < kChunk / kGateBlock; blk++, col += 2 * kGateBlock / 16)
        {
            f16_t acc = {4.0f, 4.0f, 4.0f, 4.0f, 4.0f, 4.0f, 4.0f, 4.0f, 4.0f, 4.0f, 4.0f, 4.0f, 4.0f, 4.0f, 4.0f, 4.0f};
#pragma unroll
            for (int kb = 0; kb < 4; kb++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(gw[kb], halves_to_f16(col[2 * kb]), acc, 0, 0, 0);
            // acc >= 0: the position passes.  Sign bits of the sixteen results, register 15 first, so that register i ends at bit i
            typedef uint32_t u16x_t __attribute__((ext_vector_type(16)));
            const u16x_t     bits = __builtin_bit_cast(u16x_t, acc);
            uint32_t         m    = 0;
#pragma unroll
            for (int i = 15; i >= 0; i--) m = __builtin_amdgcn_alignbit(m, bits[i], 31);
            gatemap[(kGateBlock / 16) * blk + 2 * (lane & 31) + (lane >> 5)] = (uint16_t)~m;
        }
    }
    wave_lds_fence();
    // a lane's 64 positions: 64 lane .. 64 lane + 63 (eight whole groups of 8: a run of gate survivors never leaves a lane)
    uint64_t surv = *reinterpret_cast<const uint64_t*>(&gatemap[4 * lane]);
    // the rest of the chunk is short dependent chains: raised priority lets them through the vector-dense phases of the other waves (see
    // scan1090_kernel; 0.631 -> 0.621 ms per GiB here)
    __builtin_amdgcn_s_setprio(1);
    if (npos < (uint32_t)kChunk)
    {
        const int nvalid = (int)npos - 64 * lane;
        surv &= nvalid >= 64 ? ~0ull : (nvalid <= 0 ? 0ull : ((1ull << nvalid) - 1ull));
    }

    // ---------------- prefilter survivors -> exact gate -> queue -> demodulation, kQueue24 prefilter survivors per pass
    if constexpr (diag::kParts == 1)
    { // measurement build: image + prefilter only
        Emit e  = begin_chunk(a, me);
        e.count = (surv == 0x123456789ull) ? 1 : 0; // keeps the prefilter alive
        pend    = finish_chunk(me, e);
        if (next == kNoChunk) break;
        chunk = next;
        next  = take_next(a, wr, ticket, lane);
        continue;
    }
    const uint32_t mine = (uint32_t)__builtin_popcountll(surv);
    const uint32_t incl = wave_incl_scan_add(mine);
    const uint32_t n1   = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    Emit           e = begin_chunk(a, me);
    // A pass takes whole lanes (a lane's survivors of one 8-position group are consecutive queue entries in ascending position, so
    // a run never straddles two passes) until kQueue24 entries are full.
    const uint32_t excl = incl - mine;
    // The prefilter's survivors go through the exact gate kQueue24 at a time; what passes collects in `queue` (whole groups of 8 positions:
    // a pass takes whole lanes) and is worked on -- scores, run rule, slicing -- once a further pass's survivors do not fit any more (that pass
    // is then simply run again for the next round), or at the end: once per chunk unless frames stand back to back.
    for (uint32_t base = 0; base < n1;)
    {
        uint32_t nq = 0;
        do
        {
        const uint64_t over = ballot(excl >= base && incl > base + (uint32_t)kQueue24);
        uint32_t       next_base = n1;
        int            last_lane = 64; // lanes base-lane .. last_lane - 1 are in
        if (over)
        {
            last_lane = __builtin_ctzll(over);
            next_base = (uint32_t)__builtin_amdgcn_readlane((int)excl, last_lane);
        }
        const uint32_t ng = next_base - base; // <= kQueue24; > 0 because a lane holds at most 64
        if (excl >= base && lane < last_lane)
        { // the two 32-bit halves one after the other: find-first-set and clear-lowest stay 32-bit operations
            uint32_t idx = excl - base;
#pragma unroll
            for (int h = 0; h < 2; h++)
            {
                uint32_t sv = (uint32_t)(surv >> (32 * h));
                while (sv)
                {
                    const uint32_t bit = (uint32_t)__builtin_ctz(sv);
                    sv &= sv - 1u;
                    gqueue[idx++] = (uint16_t)(64u * (uint32_t)lane + 32u * (uint32_t)h + bit);
                }
            }
        }
        wave_lds_fence();
        // ---- the exact gate (oracle2400_gate), a lane per prefilter survivor: every sum in 32 bits, the saturations of the specification
        // reduce to "2 Q below 65535" (3 lo saturated exceeds 2 Q saturated only when 2 Q is not, and then 3 lo > 2 Q decides)
        {
            const bool     in  = (uint32_t)lane < ng;
            const uint32_t pos_in = in ? gqueue[lane] : 0u;
            const uint16_t* w  = &img16[(uint32_t)kImgPad + pos_in]; // sample a of the position: w[a]
            const uint32_t A = (uint32_t)w[0] + w[1], B = (uint32_t)w[2] + w[3], C = (uint32_t)w[8] + w[9], D = (uint32_t)w[10] + w[11] + w[12];
            const uint32_t Q = (uint32_t)w[-1] + w[5] + w[6] + w[7] + w[14] + w[15] + w[16] + w[17];
            const uint32_t lo3 = 3u * __builtin_elementwise_min(__builtin_elementwise_min(A, B), __builtin_elementwise_min(C, D));
            const uint64_t okm   = ballot(in && 2u * Q < 65535u && lo3 > 2u * Q);
            const uint32_t npass = (uint32_t)__builtin_popcountll(okm);
            if (nq != 0 && nq + npass > (uint32_t)kQueue24) break; // no room: the queue is worked on first, this pass comes again
            if ((okm >> lane) & 1ull) queue[nq + (uint32_t)__builtin_popcountll(okm & ((1ull << lane) - 1ull))] = (uint16_t)pos_in;
            nq += npass;
            base = next_base;
        }
        wave_lds_fence();
        } while (base < n1);
        if constexpr (diag::kParts == 2) nq = nq == 0x12345678u ? 1u : 0u; // measurement build: + the exact gate
        // ---- preamble scores, sixteen survivors per matrix product (see kScoreTable).  First the magnitudes, four survivors per trip: row rw of 16
        // lanes takes entry t + rw, lane tl of the row the exact magnitude of its window sample tl, and stores it, bytes XOR 0x80, as half tl of
        // the survivor's 32 bytes (half 13: the constant, halves 14 and 15 carry no weight).  The buffer lives where the survivor map was.
        for (uint32_t q0 = 0; q0 < nq; q0 += 16)
        {
            const uint32_t nhere = nq - q0 < 16u ? nq - q0 : 16u;
            for (uint32_t t = 0; t < nhere; t += 4)
            {
                const uint32_t q   = q0 + t + (uint32_t)rw;
                const uint32_t pos = queue[q < nq ? q : nq - 1];
                const uint32_t m   = (uint32_t)mag_of_s(img16[(uint32_t)kImgPad + pos + (uint32_t)tl]);
                scorebuf[16u * (t + (uint32_t)rw) + (uint32_t)tl] = (uint16_t)(tl < 13 ? m ^ 0x8080u : tl == 13 ? 0x8080u : 0u);
            }
            wave_lds_fence();
            // lane (n, h): bytes 16 h .. 16 h + 15 of survivor n & 15 (the upper sixteen columns repeat the lower ones; nobody reads their results)
            typedef int v4i_t __attribute__((ext_vector_type(4)));
            typedef int v16i_t __attribute__((ext_vector_type(16)));
            const uint4  bq  = *reinterpret_cast<const uint4*>(&scorebuf[16 * (lane & 15) + 8 * (lane >> 5)]);
            const v16i_t acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(score_w, v4i_t{(int)bq.x, (int)bq.y, (int)bq.z, (int)bq.w},
                                                                     v16i_t{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
            int v[7];
#pragma unroll
            for (int k = 0; k < 7; k++) v[k] = (int)(((uint32_t)acc[8 + k] << 8) + (uint32_t)acc[k]);
            int best = v[0], phi = 0;
#pragma unroll
            for (int k = 1; k < 5; k++)
                if (v[k] > best) best = v[k], phi = k;
            // qualifies (oracle2400.c): P > 0 and 8 P >= T(phi) = 5 (m0 + .. + m11) + phi (m12 - m0), what the ten slots hold
            const int total = 5 * v[5] + phi * v[6];
            if ((uint32_t)lane < nhere)
            {
                score[q0 + (uint32_t)lane] = (best > 0 && 8 * best >= total) ? ((uint32_t)best << 3) | (uint32_t)phi : 0u;
                pulse[q0 + (uint32_t)lane] = (uint32_t)(((best + total) >> 1) / 24); // pulse - quiet = P and pulse + quiet = T: A = pulse / 24 (four slots of six fifths)
            }
            wave_lds_fence(); // the next sixteen overwrite the buffer
        }
        // ---- one candidate per run: entry q stands for its run (gate survivors at consecutive positions inside one group of 8) when
        // its score is positive, larger than every earlier member's and not smaller than any later member's
        uint32_t nw = 0;
        {
            const uint32_t q   = (uint32_t)lane;
            const bool     in  = q < nq;
            const uint32_t pos = in ? queue[q] : 0u;
            const uint32_t sc  = in ? score[q] >> 3 : 0u;
            bool           win = sc > 0u, run = win; // an entry without a score cannot stand for its run: no need to look around it
#pragma unroll 1
            for (uint32_t k = 1; k < 8 && ballot(run); k++)
            { // earlier members
                run = run && k <= (pos & 7u) && q >= k && queue[q - k] == pos - k;
                if (run && (score[q - k] >> 3) >= sc) win = false, run = false;
            }
            run = win;
#pragma unroll 1
            for (uint32_t k = 1; k < 8 && ballot(run); k++)
            { // later members
                run = run && (pos & 7u) + k < 8u && q + k < nq && queue[q + k] == pos + k;
                if (run && (score[q + k] >> 3) > sc) win = false, run = false;
            }
            const uint64_t wins = ballot(win);
            if (win) wlist[(uint32_t)__builtin_popcountll(wins & ((1ull << lane) - 1ull))] = (uint8_t)q;
            nw = (uint32_t)__builtin_popcountll(wins);
        }
        wave_lds_fence();
        if constexpr (diag::kParts == 3)
            if (nw != 0x12345678u) nw = 0; // measurement build: + scores and run rule
        // ---- the candidates, one at a time with the whole wave
        for (uint32_t w = 0; w < nw; w++)
        {
            const uint32_t q        = (uint32_t)__builtin_amdgcn_readfirstlane((int)wlist[w]);
            const uint32_t pos      = (uint32_t)__builtin_amdgcn_readfirstlane((int)queue[q]);
            const uint32_t packed   = (uint32_t)__builtin_amdgcn_readfirstlane((int)score[q]);
            const int      phi_star = (int)(packed & 7u);
            const uint32_t a0       = (uint32_t)kImgPad + pos; // half of sample t = 0
            const int      amp      = (int)(uint32_t)__builtin_amdgcn_readfirstlane((int)pulse[q]);
            // (Looking at the five DF bits first and slicing the rest only for a DF that can be accepted saved 28 vector instructions per chunk
            // in the round-2 form and cost as much in LDS round trips: dropped.)
            if (slice_and_emit(img16, a0, lane, lt, e, g0 + pos, phi_star, amp)) continue;
            if constexpr (diag::kParts == 4) continue; // measurement build: + the first slice of the candidates
            if (phi_star + 1 <= 4 && slice_and_emit(img16, a0, lane, lt, e, g0 + pos, phi_star + 1, amp)) continue;
            if (phi_star - 1 >= 0) (void)slice_and_emit(img16, a0, lane, lt, e, g0 + pos, phi_star - 1, amp);
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

} // namespace

uint32_t chunks_per_buffer_2400(uint32_t buf_samples)
{
    if (buf_samples <= (uint32_t)kSpan24) return 0;
    return (buf_samples - (uint32_t)kSpan24 + (uint32_t)kChunk - 1) / (uint32_t)kChunk;
}

hipError_t launch_scan2400(const ScanArgs& a, uint32_t* total_and_overflow, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const GatherArgs* attached)
{
    if (a.total_chunks == 0) return hipMemsetAsync(total_and_overflow, 0, kStateWords * sizeof(uint32_t), stream);
    // persistent single-wave workgroups, as many as the LDS lets a CU hold (16 x 10 208 bytes), in whole (XCD, sub-range) units
    const uint32_t   grid = scan_grid(a);
    const GatherArgs ga   = attached ? *attached : GatherArgs{};
    if (start || stop) hipExtLaunchKernelGGL(scan2400_kernel, dim3(grid), dim3(64), 0, stream, start, stop, 0, a, total_and_overflow, ga); // (either may be NULL)
    else hipLaunchKernelGGL(scan2400_kernel, dim3(grid), dim3(64), 0, stream, a, total_and_overflow, ga);
    return hipGetLastError();
}

} // namespace adsb_amd
