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
//     parks s = (I-127)^2 + (Q-127)^2 in the same interleaved LDS image (dword q = s[q] | s[q + 2048] << 16), so one packed
//     operation serves two positions and "sample q + a" is dword q + a for every a;
//   * the gate runs dense and packed on that image: pair sums of the four pulse regions against the sum of eight quiet samples,
//     saturating 16-bit adds, survivors to a queue;
//   * a candidate is demodulated by the whole wave: first the preamble correlation of the five sub-sample phases, on registers
//     (each row of 16 lanes holds the 13 magnitudes once, weighted for its phase, DPP row sums) -- most gate survivors of noise
//     end there; then per phase tried lane b slices bit b and bit 64 + b from four samples with the overlap weights of its own
//     sub-sample position -- on float estimates of the magnitudes, exactly only where a decision lies inside the estimates' error
//     margin --, ballots give the message, parity is the DPP XOR reduction of per-lane table entries, the one-bit repair a
//     ballot over per-lane syndromes (shared with the 2 MS/s kernel).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <stdint.h>

#include "scan1090.h"
#include "scan_common.hip.h"

namespace adsb_amd
{
namespace
{

constexpr int kSpan24     = 292;                          // samples after j a candidate may read (oracle2400.c)
constexpr int kHalo24     = 320;                          // halo dwords of the image (>= kSpan24, a multiple of 8)
constexpr int kImgBase    = 4;                            // dword of the image's q = 0; dword 3 = (sample before the chunk | s[2047] << 16)
constexpr int kImgDwords  = kImgBase + kHalfChunk + kHalo24 + 4;
constexpr int kQueue24    = 64;                           // queue entries per pass (a lane holds at most 64)

__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_min(as_pk(a), as_pk(b))); }
// the same, opaque to the compiler: min(x, 1) per half written in C turns into two compares, two selects and a permute
__device__ __forceinline__ uint32_t pk_min_asm(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

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
__constant__ PreambleWeights kPreamble = make_preamble_weights();

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
    for (int t = 0; t < 4; t++) w.s[t] = img16[a0 + 2 * (i0 + t)];
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
            const uint64_t ma = is_long ? ballot(syn == lt.syn_a) : ballot(lane < 56 && syn == lt.syn_s);
            const uint64_t mb = is_long ? ballot(has_b && syn == lt.syn_b) : 0ull;
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

// inclusive sums inside each row of 16 lanes (lane 15 of the row ends with the row's total); the additions carry the DPP modifier
// themselves (see wave_incl_scan_add, scan_common.hip.h): four instructions per sum.  Seven sums at once, step by step across the seven: the
// two wait states a DPP read needs after the write of its source are filled by the other chains' additions instead of by s_nops
__device__ __forceinline__ void row_scan_add7(int (&v)[7])
{
#define ROW7(step)                                                                                          \
    "v_add_u32_dpp %0, %0, %0 row_shr:" #step " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                 \
    "v_add_u32_dpp %1, %1, %1 row_shr:" #step " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                 \
    "v_add_u32_dpp %2, %2, %2 row_shr:" #step " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                 \
    "v_add_u32_dpp %3, %3, %3 row_shr:" #step " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                 \
    "v_add_u32_dpp %4, %4, %4 row_shr:" #step " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                 \
    "v_add_u32_dpp %5, %5, %5 row_shr:" #step " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                 \
    "v_add_u32_dpp %6, %6, %6 row_shr:" #step " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm("s_nop 1\n\t" ROW7(1) ROW7(2) ROW7(4) ROW7(8) "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]));
#undef ROW7
}

__global__ __launch_bounds__(64, 4) void scan2400_kernel(ScanArgs a, uint32_t* __restrict__ total_overflow)
{
    __shared__ __attribute__((aligned(16))) uint32_t img[kImgDwords];
    __shared__ uint16_t                              queue[kQueue24];
    __shared__ uint8_t                               wlist[kQueue24]; // (a queue of 64 and a byte per winner: 10 208 bytes a wave, sixteen waves a CU)
    __shared__ uint32_t                              score[kQueue24], pulse[kQueue24]; // per queue entry: best P << 3 | phase; pulse amplitude A
    uint16_t* const                                  img16 = reinterpret_cast<uint16_t*>(img);

    const int        lane = threadIdx.x;
    stamp(a.stamps, 0);
    const LaneTables lt   = load_lane_tables(a.crc_tab, lane);
    // preamble weights of this lane (see the candidate loop)
    const int tl = lane & 15, rw = lane >> 4;
    int       wpk = 0; // the five phases' weights of window sample tl, four bits each (signed, -5 .. 5)
    if (tl < 13)
        for (int k = 0; k < 5; k++) wpk |= ((int)kPreamble.p[k][tl] & 15) << (4 * k);
    if (blockIdx.x == 0 && lane < 2) total_overflow[lane] = 0; // {record total, overflow flag}: filled by the ordering pass that follows in-stream

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
    // ---------------- window -> s, the two halves of the chunk interleaved (rows j and j + 4; j = 4: row 4 again beside the halo row)
    __builtin_amdgcn_s_setprio(0);
    wave_lds_fence(); // readers of the previous chunk are done
#pragma unroll
    for (int jr = 0; jr <= kRows / 2; jr++)
    {
        const uint4 x = raw.row[jr], y = raw.row[jr + kRows / 2];
        uint32_t    t[8];
        rows_to_s2(x.x, y.x, t[0], t[1]);
        rows_to_s2(x.y, y.y, t[2], t[3]);
        rows_to_s2(x.z, y.z, t[4], t[5]);
        rows_to_s2(x.w, y.w, t[6], t[7]);
        if (jr < kRows / 2 || lane < kHalo24 / 8)
        {
            uint4* dst = reinterpret_cast<uint4*>(&img[kImgBase + jr * kRowSamples + 8 * lane]);
            dst[0]     = make_uint4(t[0], t[1], t[2], t[3]);
            dst[1]     = make_uint4(t[4], t[5], t[6], t[7]);
        }
        // the dword in front of q = 0: low half = the sample before the chunk (0 at the start of a buffer), high half = s[2047]
        if (jr == kRows / 2 - 1 && lane == 63) img16[2 * (kImgBase - 1) + 1] = (uint16_t)t[7];
    }
    if (lane == 0) img16[2 * (kImgBase - 1)] = (uint16_t)iq1_to_s(raw.front & 0xFFu, raw.front >> 8);

    // ---------------- prefetch: the next chunk's loads fly while this chunk is processed
    const uint32_t me = chunk, g0 = g.g0, npos = g.npos;
    // control traffic (the previous chunk's directory entry and sums, the ticket for the work item after `next`) in front of the loads
    publish(a, pend, lane);
    uint32_t ticket = 0;
    if (next != kNoChunk) ticket = grab_issue(a, wr, lane);
    if (next != kNoChunk)
    {
        g = chunk_geom_of(a, next, kSpan24);
        load_window<kHalo24>(g, lane, raw);
    }
    wave_lds_fence();

    // ---------------- gate (oracle2400_gate), packed: 3 min(s0+s1, s2+s3, s8+s9, s10+s11+s12) > 2 (s-1 + s5+s6+s7 + s14+s15+s16+s17)
    uint32_t surv32[2] = {0u, 0u};
#pragma unroll
    for (int b = 0; b < kHalfChunk / 512; b++)
    {
        // T[i] = dword q0 - 4 + i of the image, q0 = 512 b + 8 lane: sample a of position q0 + k is T[4 + k + a]
        // eight 16-byte reads, written out: left to itself the compiler drops the three unused leading dwords and falls back to
        // 13 misaligned 8-byte reads (lane stride 32 bytes: eight-way bank conflicts)
        uint32_t       T[32];
        const uint32_t addr = (uint32_t)(uintptr_t)&img[kImgBase + b * 512 + 8 * lane - 4];
        uint4          v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                     "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                     : "v"(addr)
                     : "memory");
        T[0] = v0.x; T[1] = v0.y; T[2] = v0.z; T[3] = v0.w; T[4] = v1.x; T[5] = v1.y; T[6] = v1.z; T[7] = v1.w;
        T[8] = v2.x; T[9] = v2.y; T[10] = v2.z; T[11] = v2.w; T[12] = v3.x; T[13] = v3.y; T[14] = v3.z; T[15] = v3.w;
        T[16] = v4.x; T[17] = v4.y; T[18] = v4.z; T[19] = v4.w; T[20] = v5.x; T[21] = v5.y; T[22] = v5.z; T[23] = v5.w;
        T[24] = v6.x; T[25] = v6.y; T[26] = v6.z; T[27] = v6.w; T[28] = v7.x; T[29] = v7.y; T[30] = v7.z; T[31] = v7.w;
        uint32_t P2[28]; // saturating pair sums s_a + s_a+1 (P2[z .. z + 16] for z = 4 .. 11)
#pragma unroll
        for (int i = 4; i < 28; i++) P2[i] = pk_add_sat(T[i], T[i + 1]);
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const int      z  = 4 + k; // T index of s0
            const uint32_t A  = P2[z], B = P2[z + 2], C = P2[z + 8], D = pk_add_sat(P2[z + 10], T[z + 12]);
            uint32_t       q  = pk_add_sat(T[z - 1], P2[z + 5]);
            q                 = pk_add_sat(q, T[z + 7]);
            q                 = pk_add_sat(q, P2[z + 14]);
            q                 = pk_add_sat(q, P2[z + 16]);
            const uint32_t lo = pk_min_u(pk_min_u(A, B), pk_min_u(C, D));
            uint32_t lo3; // 3 lo, saturating: one multiply-add instead of two additions
            asm("v_pk_mad_u16 %0, %1, %2, 0 clamp" : "=v"(lo3) : "v"(lo), "s"(0x00030003u));
            const uint32_t ok = pk_min_asm(pk_sub_sat(lo3, pk_add_sat(q, q)), 0x00010001u); // 1 per half where 3 lo > 2 q
            const u16x2    wt = {(unsigned short)(1u << k), (unsigned short)(256u << k)};
            acc               = __builtin_amdgcn_udot2(as_pk(ok), wt, acc, false);
        }
        if (b & 1) surv32[b >> 1] |= acc << 16;
        else surv32[b >> 1] = acc;
    }
    uint64_t surv = (uint64_t)surv32[0] | ((uint64_t)surv32[1] << 32);
    // the rest of the chunk is short dependent chains: raised priority lets them through the vector-dense phases of the other waves (see
    // scan1090_kernel; 0.631 -> 0.621 ms per GiB here)
    __builtin_amdgcn_s_setprio(1);
    if (npos < (uint32_t)kChunk)
    {
#pragma unroll
        for (int m = 0; m < 64; m += 8)
        {
            const int nvalid = (int)npos - (kHalfChunk * ((m >> 3) & 1) + 512 * (m >> 4) + 8 * lane);
            uint64_t  keep   = (nvalid >= 8) ? 0xFFull : (nvalid <= 0 ? 0ull : ((1ull << nvalid) - 1ull));
            surv &= ~(0xFFull << m) | (keep << m);
        }
    }

    // ---------------- candidates -> queue -> demodulation, kQueue24 per pass
    if constexpr (diag::kParts == 1)
    { // measurement build: image + gate only
        Emit e  = begin_chunk(a, me);
        e.count = (surv == 0x123456789ull) ? 1 : 0; // keeps the gate alive
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
    for (uint32_t base = 0; base < n1;)
    {
        const uint64_t over = ballot(excl >= base && incl > base + (uint32_t)kQueue24);
        uint32_t       next_base = n1;
        int            last_lane = 64; // lanes base-lane .. last_lane - 1 are in
        if (over)
        {
            last_lane = __builtin_ctzll(over);
            next_base = (uint32_t)__builtin_amdgcn_readlane((int)excl, last_lane);
        }
        const uint32_t nq = next_base - base; // <= kQueue24; > 0 because a lane holds at most 64
        if (excl >= base && lane < last_lane)
        {
            uint64_t sv  = surv;
            uint32_t idx = excl - base;
            while (sv)
            {
                const int bit = __builtin_ctzll(sv);
                sv &= sv - 1;
                queue[idx++] = (uint16_t)(kHalfChunk * ((bit >> 3) & 1) + 512 * (bit >> 4) + 8 * lane + (bit & 7));
            }
        }
        wave_lds_fence();
        // ---- preamble scores, four candidates per trip: row r of 16 lanes takes entry t + r, lane 16 r + i holds sample i of its window
        // (13 of them matter); five weighted row sums give P(phi), lane 15 of the row keeps the best (first of equals) and stores
        // best << 3 | phi, or 0 when the survivor does not qualify (P <= 0, or the pulse slots do not stand out of the ten: most gate
        // survivors of noise).
        for (uint32_t t = 0; t < nq; t += 4)
        {
            const uint32_t q   = t + (uint32_t)rw;
            const uint32_t pos = queue[q < nq ? q : nq - 1];
            const uint32_t a0  = 2u * (uint32_t)kImgBase + 2u * (pos & (uint32_t)(kHalfChunk - 1)) + (pos >> 11);
            const int      m   = mag_of_s(img16[a0 + 2 * tl]);
            // P(phi) for the five phases, m0 + .. + m11 and m12 - m0: seven sums over the row
            int v[7];
#pragma unroll
            for (int k = 0; k < 5; k++) v[k] = __builtin_amdgcn_sbfe(wpk, 4 * k, 4) * m;
            v[5] = tl < 12 ? m : 0, v[6] = tl == 12 ? m : tl == 0 ? -m : 0;
            row_scan_add7(v);
            int best = v[0], phi = 0;
#pragma unroll
            for (int k = 1; k < 5; k++)
                if (v[k] > best) best = v[k], phi = k;
            // qualifies (oracle2400.c): P > 0 and 8 P >= T(phi) = 5 (m0 + .. + m11) + phi (m12 - m0), what the ten slots hold
            const int s12 = v[5], d12 = v[6];
            const int total = 5 * s12 + phi * d12;
            if (tl == 15 && q < nq)
            {
                score[q] = (best > 0 && 8 * best >= total) ? ((uint32_t)best << 3) | (uint32_t)phi : 0u;
                pulse[q] = (uint32_t)(((best + total) >> 1) / 24); // pulse - quiet = P and pulse + quiet = T: A = pulse / 24 (four slots of six fifths)
            }
        }
        wave_lds_fence();
        // ---- one candidate per run: entry q stands for its run (gate survivors at consecutive positions inside one group of 8) when
        // its score is positive, larger than every earlier member's and not smaller than any later member's
        uint32_t nw = 0;
        for (uint32_t r0 = 0; r0 < nq; r0 += 64)
        {
            const uint32_t q   = r0 + (uint32_t)lane;
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
            if (win) wlist[nw + (uint32_t)__builtin_popcountll(wins & ((1ull << lane) - 1ull))] = (uint8_t)q;
            nw += (uint32_t)__builtin_popcountll(wins);
        }
        wave_lds_fence();
        if constexpr (diag::kParts == 2)
            if (nw != 0x12345678u) nw = 0; // measurement build: + queue, scores and run rule
        // ---- the candidates, one at a time with the whole wave
        for (uint32_t w = 0; w < nw; w++)
        {
            const uint32_t q        = (uint32_t)__builtin_amdgcn_readfirstlane((int)wlist[w]);
            const uint32_t pos      = (uint32_t)__builtin_amdgcn_readfirstlane((int)queue[q]);
            const uint32_t packed   = (uint32_t)__builtin_amdgcn_readfirstlane((int)score[q]);
            const int      phi_star = (int)(packed & 7u);
            const uint32_t a0       = 2u * (uint32_t)kImgBase + 2u * (pos & (uint32_t)(kHalfChunk - 1)) + (pos >> 11); // half of sample t = 0
            const int      amp      = (int)(uint32_t)__builtin_amdgcn_readfirstlane((int)pulse[q]);
            // (Looking at the five DF bits first and slicing the rest only for a DF that can be accepted saved 28 vector instructions per chunk
            // in the round-2 form and cost as much in LDS round trips: dropped.)
            if (slice_and_emit(img16, a0, lane, lt, e, g0 + pos, phi_star, amp)) continue;
            if constexpr (diag::kParts == 3) continue; // measurement build: + the first slice of the candidates
            if (phi_star + 1 <= 4 && slice_and_emit(img16, a0, lane, lt, e, g0 + pos, phi_star + 1, amp)) continue;
            if (phi_star - 1 >= 0) (void)slice_and_emit(img16, a0, lane, lt, e, g0 + pos, phi_star - 1, amp);
        }
        wave_lds_fence();
        base = next_base;
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

hipError_t launch_scan2400(const ScanArgs& a, uint32_t* total_and_overflow, hipStream_t stream, hipEvent_t start, hipEvent_t stop)
{
    if (a.total_chunks == 0) return hipMemsetAsync(total_and_overflow, 0, 2 * sizeof(uint32_t), stream);
    // persistent single-wave workgroups, as many as the LDS lets a CU hold (16 x 10 208 bytes), in whole (XCD, sub-range) units
    const uint32_t grid = scan_grid(a);
    if (start || stop) hipExtLaunchKernelGGL(scan2400_kernel, dim3(grid), dim3(64), 0, stream, start, stop, 0, a, total_and_overflow); // (either may be NULL)
    else hipLaunchKernelGGL(scan2400_kernel, dim3(grid), dim3(64), 0, stream, a, total_and_overflow);
    return hipGetLastError();
}

} // namespace adsb_amd
