// scan2400.hip -- gfx950 kernel of the 2.4 MS/s Mode S scan mode (ADSB_AMD_MODE_2400).
//
// Not a counterpart of anything libadsb compiles: the reference demodulates 2 samples per microsecond only (ADSB1090.cpp:148,
// 757-758).  BASELINE.json quotes its metric on "2.4 MS/s u8 IQ", so the library has this second mode; its
// definition is oracle/oracle2400.c (read that header for the signal geometry and every rule), and this kernel has to produce
// exactly the records that file produces -- parity unpinned, GPU == specification, generator -> decoder round trips.
//
// Shape, MI355X-first like the 2 MS/s kernel but not yet tuned (one wave per chunk, no persistent waves, no register prefetch):
//   * a wave owns 4096 preamble positions of one buffer, loads the 4416 samples they can touch with 16-byte coalesced loads and
//     parks s = (I-127)^2 + (Q-127)^2 in the same interleaved LDS image (dword q = s[q] | s[q + 2048] << 16), so one packed
//     operation serves two positions and "sample q + a" is dword q + a for every a;
//   * the gate runs dense and packed on that image: pair sums of the four pulse regions against the sum of eight quiet samples,
//     saturating 16-bit adds, survivors to a queue;
//   * a candidate is demodulated by the whole wave: exact magnitudes of its 292-sample window once into LDS, the preamble
//     correlation of the five sub-sample phases on five lanes, then per phase tried lane b slices bit b and bit 64 + b from four
//     magnitudes with the overlap weights of its own sub-sample position, ballots give the message, parity is the DPP XOR
//     reduction of per-lane table entries, the one-bit repair a ballot over per-lane syndromes (shared with the 2 MS/s kernel).
#include <hip/hip_runtime.h>

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
constexpr int kQueue24    = 256;
constexpr int kWinSamples = 296;                          // magnitudes kept per candidate (t = 0 .. 295, 292 used)

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

// 16 bytes of IQ at sample g of the buffer (g a multiple of 8); samples at or beyond n read as I = Q = 127 (s = 0)
__device__ __forceinline__ uint4 load_iq16(const uint8_t* __restrict__ buf, uint32_t g, uint32_t n)
{
    if (g + 8u <= n) return *reinterpret_cast<const uint4*>(buf + 2ull * g);
    uint32_t w[4] = {0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu};
    for (uint32_t k = 0; k < 8u; k++)
        if (g + k < n)
        {
            const uint32_t v  = (uint32_t)buf[2ull * (g + k)] | ((uint32_t)buf[2ull * (g + k) + 1] << 8);
            const uint32_t sh = 16u * (k & 1u);
            w[k >> 1]         = (w[k >> 1] & ~(0xFFFFu << sh)) | (v << sh);
        }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// overlap, in fifths of a sample, of the half-microsecond slot k of a frame that starts phi fifths into its first sample with
// sample t of the window (oracle2400.c: overlap5 / slot_energy)
constexpr int slot_overlap(int phi, int k, int t)
{
    const int lo = phi + 6 * k, hi = lo + 6, a = lo > 5 * t ? lo : 5 * t, b = hi < 5 * t + 5 ? hi : 5 * t + 5;
    return b > a ? b - a : 0;
}
// per phase and sample: weight of the sample in P(phi) (pulse slots 0, 2, 7, 9 minus quiet slots 1, 3, 4, 5, 6, 8) and in the
// pulse energy alone.  13 samples cover every slot up to 9 for every phase.
struct PreambleWeights
{
    int8_t p[5][13], e[5][13];
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
            w.e[phi][t] = (int8_t)pulse;
        }
    return w;
}
__constant__ PreambleWeights kPreamble = make_preamble_weights();

// One slice of the window at phase phi (wave-uniform): records what oracle2400.c's slice_phase accepts.  Returns true when a
// record was emitted.
__device__ __forceinline__ bool slice_and_emit(const uint16_t* mwin, int lane, const LaneTables& lt, Emit& e, uint32_t j, int phi, int amp)
{
    // bit b = lane (A) and 64 + lane (B, lanes < 48): 5 i0 + p = phi + 96 + 12 b; weights of phase p: {5-p, 2p-3, -min(2+p,5), -(p==4)}
    int  c[2];
    bool has_b = lane < 48;
#pragma unroll
    for (int h = 0; h < 2; h++)
    {
        const int b  = lane + 64 * h;
        const int T  = phi + 96 + 12 * (h == 0 || has_b ? b : lane);
        const int i0 = T / 5, p = T - 5 * i0;
        const int w2 = (2 + p < 5) ? 2 + p : 5;
        c[h] = (5 - p) * (int)mwin[i0] + (2 * p - 3) * (int)mwin[i0 + 1] - w2 * (int)mwin[i0 + 2] - (p == 4 ? (int)mwin[i0 + 3] : 0);
    }
    const uint64_t valA = ballot(c[0] > 0), valB = ballot(has_b && c[1] > 0);
    const uint64_t wkA = ballot(2 * abs(c[0]) < amp), wkB = ballot(has_b && 2 * abs(c[1]) < amp);
    const uint32_t df      = (uint32_t)(__builtin_bitreverse64(valA) >> 59);
    const bool     is_long = df_is_long(df);
    const uint32_t nbits   = is_long ? 112u : 56u;
    const uint64_t ba = is_long ? valA : (valA & kMask56), bb = is_long ? valB : 0ull;
    const int      weak = is_long ? __builtin_popcountll(wkA) + __builtin_popcountll(wkB) : __builtin_popcountll(wkA & kMask56);
    if (weak > (int)nbits / 8) return false;
    const bool is17 = (df == 11 || df == 17);
    if (!is17 && !df_is_ap(df)) return false;
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

__global__ __launch_bounds__(64) void scan2400_kernel(ScanArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t img[kImgDwords];
    __shared__ uint16_t                              mwin[kWinSamples + 8];
    __shared__ uint16_t                              queue[kQueue24];
    uint16_t* const                                  img16 = reinterpret_cast<uint16_t*>(img);

    const int        lane = threadIdx.x;
    const LaneTables lt   = load_lane_tables(a.crc_tab, lane);
    const uint32_t   me   = blockIdx.x;
    if (me >= a.total_chunks) return;
    const uint32_t bidx = me / a.chunks_per_buf, cidx = me % a.chunks_per_buf;
    const uint8_t* buf  = a.iq + (uint64_t)bidx * a.buf_stride;
    const uint32_t n    = a.buf_samples;
    const uint32_t g0   = cidx * (uint32_t)kChunk;
    const uint32_t lim  = n - (uint32_t)kSpan24; // positions j < lim
    const uint32_t npos = (lim - g0 < (uint32_t)kChunk) ? (lim - g0) : (uint32_t)kChunk;

    // ---------------- window -> s, the two halves of the chunk interleaved (rows j and j + 4; j = 4: row 4 again beside the halo row)
#pragma unroll
    for (int jr = 0; jr <= kRows / 2; jr++)
    {
        if (jr == kRows / 2 && lane >= kHalo24 / 8) break;
        const uint4 x = load_iq16(buf, g0 + (uint32_t)(jr * kRowSamples + 8 * lane), n);
        const uint4 y = load_iq16(buf, g0 + (uint32_t)((jr + kRows / 2) * kRowSamples + 8 * lane), n);
        uint32_t    t[8];
        rows_to_s2(x.x, y.x, t[0], t[1]);
        rows_to_s2(x.y, y.y, t[2], t[3]);
        rows_to_s2(x.z, y.z, t[4], t[5]);
        rows_to_s2(x.w, y.w, t[6], t[7]);
        uint4* dst = reinterpret_cast<uint4*>(&img[kImgBase + jr * kRowSamples + 8 * lane]);
        dst[0]     = make_uint4(t[0], t[1], t[2], t[3]);
        dst[1]     = make_uint4(t[4], t[5], t[6], t[7]);
        // the dword in front of q = 0: low half = the sample before the chunk (0 at the start of a buffer), high half = s[2047]
        if (jr == kRows / 2 - 1 && lane == 63) img16[2 * (kImgBase - 1) + 1] = (uint16_t)t[7];
    }
    if (lane == 0)
    {
        uint32_t f = 0;
        if (g0 > 0) f = iq1_to_s(buf[2ull * (g0 - 1)], buf[2ull * (g0 - 1) + 1]);
        img16[2 * (kImgBase - 1)] = (uint16_t)f;
    }
    wave_lds_fence();

    // ---------------- gate (oracle2400_gate), packed: 2 min(s0+s1, s2+s3, s8+s9, s10+s11+s12) > s-1 + s5+s6+s7 + s14+s15+s16+s17
    uint32_t surv32[2] = {0u, 0u};
#pragma unroll 1
    for (int b = 0; b < kHalfChunk / 512; b++)
    {
        // T[i] = dword q0 - 4 + i of the image, q0 = 512 b + 8 lane: sample a of position q0 + k is T[4 + k + a]
        uint32_t        T[32];
        const uint32_t* p = &img[kImgBase + b * 512 + 8 * lane - 4];
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            const uint4 v = *reinterpret_cast<const uint4*>(p + 4 * i);
            T[4 * i] = v.x; T[4 * i + 1] = v.y; T[4 * i + 2] = v.z; T[4 * i + 3] = v.w;
        }
        uint32_t P2[29]; // saturating pair sums s_a + s_a+1
#pragma unroll
        for (int i = 3; i < 29; i++) P2[i] = pk_add_sat(T[i], T[i + 1]);
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
            const uint32_t ok = pk_min_u(pk_sub_sat(pk_add_sat(lo, lo), q), 0x00010001u); // 1 per half where 2 lo > q
            const u16x2    wt = {(unsigned short)(1u << k), (unsigned short)(256u << k)};
            acc               = __builtin_amdgcn_udot2(as_pk(ok), wt, acc, false);
        }
        if (b & 1) surv32[b >> 1] |= acc << 16;
        else surv32[b >> 1] = acc;
    }
    uint64_t surv = (uint64_t)surv32[0] | ((uint64_t)surv32[1] << 32);
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
    const uint32_t mine = (uint32_t)__builtin_popcountll(surv);
    const uint32_t incl = wave_incl_scan_add(mine);
    const uint32_t n1   = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    Emit           e;
    e.base  = reinterpret_cast<uint4*>(a.chunk_records + (uint64_t)me * a.cap);
    e.cap   = a.cap;
    e.count = 0;
    for (uint32_t base = 0; base < n1; base += (uint32_t)kQueue24)
    {
        {
            uint64_t sv  = surv;
            uint32_t idx = incl - mine;
            while (sv)
            {
                const int bit = __builtin_ctzll(sv);
                sv &= sv - 1;
                if (idx - base < (uint32_t)kQueue24) queue[idx - base] = (uint16_t)(kHalfChunk * ((bit >> 3) & 1) + 512 * (bit >> 4) + 8 * lane + (bit & 7));
                idx++;
            }
        }
        wave_lds_fence();
        const uint32_t nq = (n1 - base < (uint32_t)kQueue24) ? (n1 - base) : (uint32_t)kQueue24;
        for (uint32_t t = 0; t < nq; t++)
        {
            const uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)queue[t]);
            const uint32_t a0  = 2u * (uint32_t)kImgBase + 2u * (pos & (uint32_t)(kHalfChunk - 1)) + (pos >> 11); // half of sample t = 0
            // exact magnitudes of the window, once
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < 5; i++)
            {
                const int tt = lane + 64 * i;
                if (tt < kWinSamples) mwin[tt] = (uint16_t)mag_of_s(img16[a0 + 2 * tt]);
            }
            wave_lds_fence();
            // preamble correlation of the five phases, one per lane
            int P = 0, E = 0;
            if (lane < 5)
            {
#pragma unroll
                for (int tt = 0; tt < 13; tt++)
                {
                    const int m = (int)mwin[tt];
                    P += (int)kPreamble.p[lane][tt] * m;
                    E += (int)kPreamble.e[lane][tt] * m;
                }
            }
            int best = __builtin_amdgcn_readlane(P, 0), phi_star = 0;
#pragma unroll
            for (int phi = 1; phi < 5; phi++)
            {
                const int v = __builtin_amdgcn_readlane(P, phi);
                if (v > best) best = v, phi_star = phi;
            }
            if (best <= 0) continue;
            int amp = 0;
#pragma unroll
            for (int phi = 0; phi < 5; phi++)
            {
                const int v = __builtin_amdgcn_readlane(E, phi);
                if (phi == phi_star) amp = v / 24;
            }
            if (slice_and_emit(mwin, lane, lt, e, g0 + pos, phi_star, amp)) continue;
            if (phi_star + 1 <= 4 && slice_and_emit(mwin, lane, lt, e, g0 + pos, phi_star + 1, amp)) continue;
            if (phi_star - 1 >= 0) (void)slice_and_emit(mwin, lane, lt, e, g0 + pos, phi_star - 1, amp);
        }
        wave_lds_fence();
    }
    if (lane == 0) a.chunk_counts[me] = e.count;
}

} // namespace

uint32_t chunks_per_buffer_2400(uint32_t buf_samples)
{
    if (buf_samples <= (uint32_t)kSpan24) return 0;
    return (buf_samples - (uint32_t)kSpan24 + (uint32_t)kChunk - 1) / (uint32_t)kChunk;
}

hipError_t launch_scan2400(const ScanArgs& a, uint32_t* total_and_overflow, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(total_and_overflow, 0, 2 * sizeof(uint32_t), stream);
    if (e != hipSuccess || a.total_chunks == 0) return e;
    hipLaunchKernelGGL(scan2400_kernel, dim3(a.total_chunks), dim3(64), 0, stream, a);
    return hipGetLastError();
}

} // namespace adsb_amd
