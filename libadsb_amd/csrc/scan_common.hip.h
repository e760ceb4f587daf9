// scan_common.hip.h -- device helpers shared by the gfx950 scan kernels (scan1090.hip: the reference's 2 samples per microsecond;
// scan2400.hip: the 2.4 MS/s mode): packed 16-bit arithmetic, u8 IQ -> s = (I-127)^2 + (Q-127)^2, the exact reference magnitude,
// wave64 reductions on DPP, per-lane parity tables, and the raw record a demodulating wave emits.
#pragma once

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>

#include "diag.hip.h"
#include "scan1090.h"

namespace adsb_amd
{
namespace
{

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short          i16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2    as_pk(uint32_t x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ uint32_t as_u32(u16x2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_max(as_pk(a), as_pk(b))); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u32(as_pk(a) - as_pk(b)); }

// Two IQ samples (4 bytes I0 Q0 I1 Q1) -> (s0 | s1 << 16), s = (I-127)^2 + (Q-127)^2 saturated to 32767.
// 32768 (I = Q = 255) is the only value above 32767 and 32767 itself is not a sum of two squares, so the
// saturation keeps the order of all reachable values and makes every difference fit in an int16.
// Six VALU ops per pair of samples: two byte permutes unpack I and Q into 16-bit halves, two packed subtracts remove the
// 127 offset, one packed multiply and one saturating packed multiply-add (signed clamp = 32767) finish.
__device__ __forceinline__ uint32_t iq2_to_s2(uint32_t x)
{
    const uint32_t i16 = __builtin_amdgcn_perm(0u, x, 0x0C020C00u); // (I0, I1): bytes 0 and 2, zero-extended (selector 0x0C = 0x00)
    const uint32_t q16 = __builtin_amdgcn_perm(0u, x, 0x0C030C01u); // (Q0, Q1): bytes 1 and 3
    const u16x2    c   = {127, 127};
    const uint32_t di  = as_u32(as_pk(i16) - c), dq = as_u32(as_pk(q16) - c);
    const uint32_t a   = as_u32(as_pk(di) * as_pk(di)); // (I-127)^2 <= 16384, exact modulo 2^16
    uint32_t       r;
    asm("v_pk_mad_i16 %0, %1, %1, %2 clamp" : "=v"(r) : "v"(dq), "v"(a));
    return r;
}

// (127 - I) and (127 - Q) of two samples as sign-extended 16-bit pairs -> (s_a | s_b << 16), s saturated to 32767 as in iq2_to_s2
__device__ __forceinline__ uint32_t pair_to_s2(uint32_t di, uint32_t dq)
{
    const uint32_t a = as_u32(as_pk(di) * as_pk(di)); // (127-I)^2 <= 16384, exact modulo 2^16
    uint32_t       r;
    asm("v_pk_mad_i16 %0, %1, %1, %2 clamp" : "=v"(r) : "v"(dq), "v"(a));
    return r;
}

// The same 4 bytes (two samples) of two rows, x from the lower half of the chunk and y from the upper -> two dwords of the image:
// t0 = (s(x sample 0), s(y sample 0)), t1 = (s(x sample 1), s(y sample 1)).
// Byte ^ 0x7F is 127 - byte as a signed 8-bit number, for every byte value (127 - 255 = -128 included), so one XOR removes the
// offset of all four bytes of a register and squaring does not care about the sign.  v_perm_b32 then builds the 16-bit pairs:
// selector 0..3 = bytes of the second operand, 4..7 = bytes of the first, and 8 / 9 / 10 / 11 replicate the sign bit of byte
// 1 / 3 / 5 / 7, i.e. it sign-extends the odd bytes (Q) directly; for the even bytes (I) the registers are first shifted up
// by one byte.  Per dword of the image: 4 half-rate operations (two permutes, multiply, multiply-add) and 2 full-rate ones
// (the XOR and the shift are shared by the two dwords a register pair yields), where unpack + subtract + square took 6
// half-rate ones.
__device__ __forceinline__ void rows_to_s2(uint32_t x, uint32_t y, uint32_t& t0, uint32_t& t1)
{
    const uint32_t zx = x ^ 0x7F7F7F7Fu, zy = y ^ 0x7F7F7F7Fu;
    const uint32_t ux = zx << 8, uy = zy << 8;
    t0 = pair_to_s2(__builtin_amdgcn_perm(uy, ux, 0x0A050801u), __builtin_amdgcn_perm(zy, zx, 0x0A050801u));
    t1 = pair_to_s2(__builtin_amdgcn_perm(uy, ux, 0x0B070903u), __builtin_amdgcn_perm(zy, zx, 0x0B070903u));
}

__device__ __forceinline__ uint32_t iq1_to_s(uint32_t i, uint32_t q)
{
    int      di = (int)i - 127, dq = (int)q - 127;
    uint32_t s  = (uint32_t)(di * di + dq * dq);
    return s > 32767u ? 32767u : s;
}

// Exact reference magnitude round(sqrt(s) * 360) (ADSB1090.cpp:138) from the saturated s.
__device__ __forceinline__ int mag_of_s(uint32_t s)
{
    const float    f = __builtin_amdgcn_sqrtf((float)s);
    const int      m = (int)(uint32_t)__builtin_fmaf(f, 360.0f, 0.5f);
    const uint32_t t = __umul24(s, 129600u);                  // < 2^32
    const int      d = (int)(t - (uint32_t)__umul24(m, m));   // m <= 65167: exact in 32 bits
    // 129600 s > m^2 + m: estimate one too small;  129600 s <= m^2 - m: one too large (max() keeps s = 0 at 0)
    const int      r = __builtin_elementwise_max(m + (d > m ? 1 : 0) - (d + m <= 0 ? 1 : 0), 0);
    return (s == 32767u) ? 65167 : r;                          // 32767 stands for s = 32768 (I = Q = 255)
}

// Float estimate of the same magnitude: |est - 360*sqrt(s)| < 0.05 (v_sqrt_f32 is good to 1 ulp, one more rounding in
// the multiply; 32767 standing for 32768 costs at most 1.0), hence |est - mag_of_s| < 1.6.
__device__ __forceinline__ float mag_estimate(uint32_t s) { return __builtin_amdgcn_sqrtf((float)s) * 360.0f; }
constexpr float kEstErr = 1.6f;

__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
// Ordering point between LDS writes and reads of other lanes of the same wavefront (single-wave workgroups): the
// hardware executes one wave's LDS instructions in order, the fence only stops the compiler from reordering them.
// Unlike __syncthreads() it does not wait for outstanding global loads, so the prefetched window stays in flight.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// wave64 scans / reductions on DPP (no LDS round trips): row_shr 1,2,4,8 inside each row of 16, then
// row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3.  Lane 63 ends up with the full result.
// ------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t x)
{
    // The additions carry the DPP modifier themselves (from the intrinsic form the compiler makes v_mov_b32_dpp + v_add_u32: twelve
    // vector instructions instead of six).  Out-of-row sources read as zero (bound_ctrl), rows masked off keep their value.  A DPP
    // operand written by the previous vector instruction needs two wait states; the compiler does not look inside the block.
    asm("s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1"
        : "+v"(x));
    return x;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_add(v), 63); }
// Two sums at once, the two chains interleaved step by step: a DPP operand written by the previous vector instruction needs two wait states, and the
// other chain's step is one of them (round 6: the general demodulation path forms its two energy sums back to back)
__device__ __forceinline__ void wave_sum2(uint32_t a, uint32_t b, uint32_t& sa, uint32_t& sb)
{
    asm("s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
    sa = (uint32_t)__builtin_amdgcn_readlane((int)a, 63), sb = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
}
__device__ __forceinline__ uint32_t wave_xor(uint32_t x)
{
    x ^= dpp_or_zero<0x111, 0xF>(x);
    x ^= dpp_or_zero<0x112, 0xF>(x);
    x ^= dpp_or_zero<0x114, 0xF>(x);
    x ^= dpp_or_zero<0x118, 0xF>(x);
    x ^= dpp_or_zero<0x142, 0xA>(x);
    x ^= dpp_or_zero<0x143, 0xC>(x);
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
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

constexpr uint64_t kMask48 = (1ull << 48) - 1ull;
constexpr uint64_t kMask56 = (1ull << 56) - 1ull;

// 112 sliced bits from per-lane decisions: bit b keeps the value of the last decided bit <= b (:838).
// Fast path (every bit decided, the normal case for a real frame): the value ballots are the message.
__device__ __forceinline__ void slice_resolve(int lane, bool has_b, bool decided_a, bool value_a, bool decided_b, bool value_b, uint64_t* ba,
                                              uint64_t* bb)
{
    const uint64_t decA = ballot(decided_a), valA = ballot(value_a);
    const uint64_t decB = ballot(has_b && decided_b), valB = ballot(has_b && value_b);
    if (decA == ~0ull && decB == kMask48)
    {
        *ba = valA;
        *bb = valB;
        return;
    }
    const uint32_t bitA = hold_resolve(decA, valA, lane, 0u);
    const uint32_t last = hold_resolve(decA, valA, 63, 0u);
    const uint32_t bitB = hold_resolve(decB, valB, lane, last);
    *ba = ballot(bitA != 0);
    *bb = ballot(has_b && bitB != 0);
}

struct LaneTables
{
    uint32_t crc_a; // parity-table entry of bit `lane` (112-bit) = the syndrome of flipping it (lane < 64 < 88: always a data bit)
    uint32_t crc_b; // parity-table entry of bit 64+lane
    uint32_t crc_s; // parity-table entry of bit `lane` of a 56-bit message
    // The syndromes of flipping bit 64 + lane of a long message / bit `lane` of a short one -- the table entry for a data bit, the bit itself inside
    // the parity field (FixSingleBitErrors :304-332) -- are worked out where a repair is tried, the one frame in ten (round 6; as three more lane
    // constants kept for the whole launch they were registers the ordering pass in front of the scan and the prefetched window can use).
    __device__ __forceinline__ uint32_t syn_b(int lane) const { return lane < 24 ? crc_b : (lane < 48 ? 1u << ((47 - lane) & 31) : 0xFFFFFFFFu); }
    __device__ __forceinline__ uint32_t syn_s(int lane) const { return lane < 32 ? crc_s : (lane < 56 ? 1u << ((55 - lane) & 31) : 0xFFFFFFFFu); }
};

__device__ __forceinline__ LaneTables load_lane_tables(const uint32_t* __restrict__ tab, int lane)
{
    LaneTables t;
    // ModesChecksumTable semantics (ADSB1090.cpp:266-275): entry b < 88 = x^(111-b) mod G, last 24 entries 0.
    uint32_t ta = tab[lane];
    uint32_t tb = (lane < 48) ? tab[64 + lane] : 0u;
    uint32_t ts = (lane < 56) ? tab[56 + lane] : 0u;
    t.crc_a     = ta;
    t.crc_b     = tb;
    t.crc_s     = ts;
    return t;
}

struct Emit
{
    uint4*   base;  // where this chunk's records go, one raw record = 2 x uint4
    uint32_t cap;   // room there
    uint32_t count; // wave-uniform
};

// A raw record is what the wave has in scalar registers anyway; turning it into the public adsb_amd_record_t (byte
// order, repair flip, address extraction) is done later by the gather kernel, one record per lane.
//   lo = { offset, df | nbits<<8 | flags<<16 | (errorbit+1)<<24, AP xor parity, reserved16 },  hi = message bits 0..127 (bit n = bit n)
// It is written with scalar stores, straight from those registers: moving the eight words into vector registers first cost eight
// vector instructions per record, in a kernel that is bound by vector issue.  Scalar stores go through the scalar data cache; every
// wave writes it back (flush_records) before it ends -- without that the ordering pass reads stale lines (tools/isa_probe.hip).
typedef uint32_t u32x4_s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void emit_raw(Emit& e, int lane, uint32_t offset, uint64_t ba, uint64_t bb, uint32_t df, uint32_t nbits,
                                         int errorbit, uint32_t flags, uint32_t syn, uint32_t extra16 = 0u)
{
    (void)lane;
    if (e.count < e.cap)
    {
        // all of it is wave-uniform; where the compiler keeps a copy in a vector register (a value that went through a vector
        // comparison) the first lane's is taken -- a scalar-register constraint on a vector value is not diagnosed
        auto           sc = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        const u32x4_s  lo = {sc(offset), sc(df | (nbits << 8) | (flags << 16) | ((uint32_t)(errorbit + 1) << 24)), sc(syn), sc(extra16)};
        const u32x4_s  hi = {sc((uint32_t)ba), sc((uint32_t)(ba >> 32)), sc((uint32_t)bb), sc((uint32_t)(bb >> 32))};
        const uint64_t p  = reinterpret_cast<uint64_t>(e.base + 2 * e.count);
        // The wait is needed: a scalar store has NOT read its data registers when it issues (tools/isa_probe.hip overwrites them right
        // after the store and finds the new values in memory), and the compiler, which cannot see into the block, reuses them freely.
        asm volatile("s_store_dwordx4 %0, %2, 0x0\n\t"
                     "s_store_dwordx4 %1, %2, 0x10\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     :
                     : "s"(lo), "s"(hi), "s"(p)
                     : "memory");
    }
    e.count++; // counts past cap signal overflow to the ordering pass
}
__device__ __forceinline__ void flush_records() { asm volatile("s_dcache_wb" ::: "memory"); }

__device__ __forceinline__ bool df_is_long(uint32_t df) { return df == 16 || df == 17 || df == 19 || df == 20 || df == 21; }
__device__ __forceinline__ bool df_is_ap(uint32_t df) { return df == 0 || df == 4 || df == 5 || df == 16 || df == 20 || df == 21 || df == 24; }

// Measurement builds only (diag.hip.h, tools/stamps.py): when every workgroup of a kernel came in and went out, on the constant
// 100 MHz clock -- a time line of the stream without a profiler attached.  One plain store per wave and event into a slot of its own
// (the host takes minimum and maximum: atomics on one word from 4096 waves cost the scan 70 us).  Empty in the product build.
constexpr uint32_t kStampGroups = 8192; // workgroups a launch may have in a measurement build
__device__ __forceinline__ void stamp(unsigned long long* st, uint32_t which)
{
    if constexpr (diag::kStamps)
    {
        const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); // every wave for itself
        if (st && (threadIdx.x & 63u) == 0 && w < kStampGroups) st[(size_t)which * kStampGroups + w] = (unsigned long long)wall_clock64();
    }
}

// ------------------------------------------------------------------------------------------------
// chunk geometry, the raw window of a chunk in registers, and the work distribution of the persistent scan kernels
// ------------------------------------------------------------------------------------------------
struct ChunkGeom
{
    const uint8_t* buf;  // reference buffer base
    uint32_t       bidx; // buffer index
    uint32_t       g0;   // first position of the chunk inside the buffer
    uint32_t       npos; // valid positions in the chunk (<= kChunk)
    uint32_t       n;    // samples in the buffer
};

// span: samples after a position a candidate there may read (positions j < n - span)
__device__ __forceinline__ ChunkGeom chunk_geom(const ScanArgs& a, uint32_t bidx, uint32_t cidx, int span)
{
    ChunkGeom g;
    g.bidx               = bidx;
    g.n                  = a.buf_samples;
    g.g0                 = cidx * (uint32_t)kChunk;
    const uint32_t limit = g.n - (uint32_t)span;
    g.npos               = (limit - g.g0 < (uint32_t)kChunk) ? (limit - g.g0) : (uint32_t)kChunk;
    g.buf                = a.iq + (uint64_t)g.bidx * a.buf_stride;
    return g;
}

// Geometry of chunk number n of the scan (wave-uniform): buffer = n / chunks_per_buf by multiplication with floor(2^32 / d), which
// is at most one too small for any n < 2^32, and one correction.  Scalar instructions only: the compiler's own expansion of the
// division keeps a float reciprocal in a vector register and then does the whole address computation of every chunk in the vector unit.
__device__ __forceinline__ ChunkGeom chunk_geom_of(const ScanArgs& a, uint32_t n, int span)
{
    n             = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    uint32_t bidx = __umulhi(n, a.cpb_magic);
    uint32_t cidx = n - bidx * a.chunks_per_buf;
    // one too small: cidx in [d, 2 d).  Written with a wrapping subtraction (d < 2^31); the empty asm statements pin the values to
    // scalar registers (otherwise the condition becomes a lane mask, the increment a v_cndmask, and everything after it vector code).
    const uint32_t t  = cidx - a.chunks_per_buf;
    uint32_t       ge = (t >> 31) ^ 1u;
    asm("" : "+s"(ge));
    bidx += ge;
    cidx = cidx < t ? cidx : t;
    asm("" : "+s"(bidx), "+s"(cidx));
    return chunk_geom(a, bidx, cidx, span);
}

// 16 bytes of IQ at sample g of the buffer; samples beyond the buffer end read as I = Q = 127 (s = 0) and are never
// used by a valid position.  Slow path, only the last chunk of a buffer comes here.
__device__ __noinline__ uint4 load_iq16_tail(const uint8_t* __restrict__ buf, uint32_t g, uint32_t n)
{
    uint32_t w[4] = {0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu};
    for (uint32_t k = 0; k < 8u; k++)
    {
        if (g + k < n)
        {
            uint32_t v  = (uint32_t)buf[2ull * (g + k)] | ((uint32_t)buf[2ull * (g + k) + 1] << 8);
            uint32_t sh = 16u * (k & 1u);
            w[k >> 1]   = (w[k >> 1] & ~(0xFFFFu << sh)) | (v << sh);
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

struct RawWindow
{
    uint4    row[kRows + 1]; // row 8 (the halo) only on lanes 0 .. HALO / 8 - 1
    uint32_t front;          // lane 0: the two bytes of sample g0-1
    // SPLIT: instead of row 8, the halo spread over all 64 lanes, four samples each: samples kChunk + 4 lane .. + 3
    uint2    cont_hi;
};

// eight bytes (four samples) at sample g of the buffer, guarded like load_iq16_tail
__device__ __forceinline__ uint2 load_iq8_guarded(const uint8_t* __restrict__ buf, uint32_t g, uint32_t n)
{
    if (g + 4u <= n) return *reinterpret_cast<const uint2*>(buf + 2ull * g);
    if (g >= n) return make_uint2(0x7F7F7F7Fu, 0x7F7F7F7Fu);
    const uint4 v = load_iq16_tail(buf, g, n);
    return make_uint2(v.x, v.y);
}

// HALO: samples after the chunk's 4096 that belong to the window (a multiple of 8, at most one row)
// FRONT: also fetch the sample in front of the chunk (lane 0)
// SPLIT (HALO == 256 only): fetch the halo as cont_hi (four samples per lane) instead of row 8
template <int HALO, bool FRONT = true, bool SPLIT = false>
__device__ __forceinline__ void load_window(const ChunkGeom& g, int lane, RawWindow& r)
{
    static_assert(!SPLIT || HALO == 4 * kLanes, "the split continuation is four samples per lane");
    if (g.g0 + (uint32_t)(kChunk + HALO) <= g.n)
    { // whole window inside the buffer (wave-uniform): plain coalesced 16-byte loads, 1 KiB per instruction.  The address is a
      // wave-uniform base plus a 32-bit lane offset that never changes, so the loads take the base from scalar registers and no
      // vector instruction is spent on addresses.  Lanes beyond the halo repeat the halo's loads (their copy is never used) instead
      // of being masked off and zeroed.
        static_assert(kRows <= 8 && kRowSamples * 2 == 1024, "the row offsets below assume 1 KiB rows, four per 4 KiB base");
        // Three scalar bases 4 KiB apart (a load's immediate offset reaches 4095) pinned to scalar registers, and explicitly global
        // pointers rebuilt from them (an integer that went through a register constraint has forgotten its address space).
        typedef const __attribute__((address_space(1))) uint8_t* gptr_t;
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        typedef const __attribute__((address_space(1))) u32x4_t* gvec_t;
        typedef const __attribute__((address_space(1))) u32x2_t* gvec2_t;
        uint64_t b0 = reinterpret_cast<uint64_t>(g.buf) + 2ull * g.g0, b1 = b0 + 4096u, b2 = b0 + 2u * (uint32_t)kChunk;
        asm("" : "+s"(b0), "+s"(b1), "+s"(b2));
        static_assert(HALO / 8 >= kLanes / 2, "one subtraction folds a lane beyond the halo back into it");
        const uint32_t off = 16u * (uint32_t)lane, off_halo = 16u * (uint32_t)(lane >= HALO / 8 ? lane - HALO / 8 : lane); // (lane modulo the halo's lanes)
#pragma unroll
        for (int k = 0; k < kRows + (SPLIT ? 0 : 1); k++)
        {
            const u32x4_t v = k < kRows ? *reinterpret_cast<gvec_t>(reinterpret_cast<gptr_t>(k < 4 ? b0 : b1) + off + (uint32_t)((k & 3) * kRowSamples * 2))
                                        : *reinterpret_cast<gvec_t>(reinterpret_cast<gptr_t>(b2) + off_halo);
            r.row[k]        = make_uint4(v.x, v.y, v.z, v.w);
        }
        if (SPLIT)
        { // 8 bytes per lane: the 256 samples behind the chunk
            const u32x2_t hi = *reinterpret_cast<gvec2_t>(reinterpret_cast<gptr_t>(b2) + 8u * (uint32_t)lane);
            r.cont_hi        = make_uint2(hi.x, hi.y);
        }
    }
    else if (g.g0 + (uint32_t)kChunk <= g.n)
    { // The buffer's last chunk when the buffer is a whole number of rows (every reference buffer is: 131072 samples): all eight rows lie inside,
      // only the halo crosses the end.  Eight plain loads back to back and a guarded halo -- the general path below waits for every row
      // before it looks at the next (a call may follow each), eight memory round trips in a row for one chunk in 32: those chunks were
      // the stragglers at the end of a launch.
        const uint8_t* const base = g.buf + 2ull * g.g0 + 16u * (uint32_t)lane;
#pragma unroll
        for (int k = 0; k < kRows; k++) r.row[k] = *reinterpret_cast<const uint4*>(base + (uint32_t)(k * kRowSamples * 2));
        const uint32_t gh = g.g0 + (uint32_t)kChunk; // first sample of the halo
        if (SPLIT) r.cont_hi = load_iq8_guarded(g.buf, gh + 4u * (uint32_t)lane, g.n);
        else
        {
            const uint32_t gk = gh + 8u * (uint32_t)lane;
            if (gk + 8u <= g.n) r.row[kRows] = *reinterpret_cast<const uint4*>(g.buf + 2ull * gk);
            else if (gk < g.n) r.row[kRows] = load_iq16_tail(g.buf, gk, g.n);
            else r.row[kRows] = make_uint4(0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu);
        }
    }
    else
    { // a chunk that the buffer's end cuts through: lanes whose 16 bytes lie inside still use the vector load, the rest the guarded path
        const uint32_t gl = g.g0 + 8u * (uint32_t)lane;
#pragma unroll
        for (int k = 0; k < kRows + (SPLIT ? 0 : 1); k++)
        {
            const uint32_t gk = gl + (uint32_t)(k * kRowSamples);
            if (gk + 8u <= g.n) r.row[k] = *reinterpret_cast<const uint4*>(g.buf + 2ull * gk);
            else if (gk < g.n) r.row[k] = load_iq16_tail(g.buf, gk, g.n);
            else r.row[k] = make_uint4(0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu);
        }
        if (SPLIT) r.cont_hi = load_iq8_guarded(g.buf, g.g0 + (uint32_t)kChunk + 4u * (uint32_t)lane, g.n);
    }
    r.front = 0x7F7Fu;
    if (FRONT && lane == 0 && g.g0 > 0) r.front = *reinterpret_cast<const uint16_t*>(g.buf + 2ull * (g.g0 - 1));
}

// Where the records of chunk `me` go: its own region of `cap` raw records.  Everything is wave-uniform.
__device__ __forceinline__ Emit begin_chunk(const ScanArgs& a, uint32_t me)
{
    Emit e;
    e.count = 0;
    e.cap   = a.cap;
    e.base  = reinterpret_cast<uint4*>(a.chunk_records + (uint64_t)me * a.cap);
    return e;
}

// What a finished chunk leaves for the ordering pass (wave-uniform): its record count (clamped to the region size) in the chunk directory, and
// the same count added to the sum of its group of kOrderChunks chunks -- the ordering pass starts from finished sums instead of running a
// summing kernel first.  The atomics need no reply; each sum sits on its own cache line (256 additions per line over the whole scan).
// It is written out (publish) half a chunk LATER, just in front of
// the next prefetch: a wave waits for its prefetched window with s_waitcnt vmcnt(0) at the top of every trip, vector-memory operations
// complete in issue order, and a store or an atomic issued at the end of a trip -- acknowledged by the memory side a microsecond or two
// later -- made that wait cover them (profiles/r04_sweep.txt).  In front of the prefetch they are older than the loads the wait is for.
struct Pending
{
    uint32_t chunk, kept;
    bool     over, valid;
};
__device__ __forceinline__ Pending finish_chunk(uint32_t chunk, const Emit& e)
{
    Pending p;
    p.chunk = chunk;
    p.kept  = e.count < e.cap ? e.count : e.cap;
    p.over  = e.count > e.cap;
    p.valid = true;
    return p;
}
__device__ __forceinline__ void publish(const ScanArgs& a, const Pending& p, int lane)
{
    // Vector atomics, one lane: they are performed where all XCDs see them.  The scalar unit's atomics (s_atomic_add) would save the six
    // vector instructions this costs, but they act on the issuing XCD's L2 only -- tools/isa_probe.hip loses additions from different XCDs
    // to one word -- and a group of kOrderChunks chunks may straddle two XCD ranges.
    if (!p.valid || lane != 0) return;
    a.chunk_dir[p.chunk] = p.kept;
    uint32_t* sum = a.block_sums + (p.chunk / kOrderChunks) * kSumStride;
    if (p.kept) atomicAdd(sum, p.kept);
    if (p.over) atomicOr(sum + 1, 1u);
}

// Chunk order.  Workgroups b and b + nxcd share an XCD (round-robin dispatch).  The chunks are dealt out in groups of 16
// consecutive chunks (one by one for inputs too small to give every counter several groups): group gg belongs to XCD gg % nxcd, and inside an XCD to work counter (gg / nxcd) % kSubRanges.  So
//   * neighbouring chunks mostly run on one XCD, back to back (a chunk's halo is the start of the next chunk: an L2 hit in 15 of 16 cases);
//   * all XCDs and all counters work on the SAME few megabytes of the input at any time: the whole chip sweeps the recording once, front
//     to back.  Giving every XCD (and every counter) its own contiguous range instead -- 32 read streams 32 MB apart -- cost 17 % of the
//     memory throughput: the loads of this kernel alone, nothing computed, took 0.201 ms per GiB against 0.171 ms for one sweep
//     (tools/membench.hip reads the same 8.5 KB pieces in one sweep; profiles/r03_sweep.txt), presumably DRAM row locality;
//   * no more than 128 waves share a counter (each on its own 128-byte line: 131 072 atomics on eight counters that shared one line took
//     1.1 ms; on 32 lines they cost nothing measurable).
// A wave's work items are numbered v = 0, 1, .. inside its counter's share ("virtual" indices): the first two are fixed (slot, slot +
// nslot), every later one comes from the counter -- the ticket drawn in front of a trip's prefetch, its value read at the end of the trip: half a
// chunk's time for the atomic (take_next) -- (chunks differ in cost -- candidates
// to demodulate --, and with a fixed stride the slowest of 4096 waves sets the kernel time).  chunk_of(v) is the chunk.
struct WorkRange
{
    uint32_t first, end; // virtual indices of this wave's counter: [0, end)
    uint32_t slot;       // index of the wave among those of the counter
    uint32_t nslot;      // waves per counter
    uint32_t range;      // which counter
    uint32_t xcd, sub, nxcd;
    uint32_t glog;       // log2 of the group size (ScanArgs::group_log2)
    uint32_t counter;    // index of the counter the wave draws from: its own (range), later a pool counter (kPoolBase + ..)
    __device__ __forceinline__ uint32_t chunk_of(uint32_t v) const
    {
        const uint32_t gi = (v >> glog) * kSubRanges + sub; // group index inside the XCD
        return (((gi * nxcd + xcd) << glog) | (v & ((1u << glog) - 1u)));
    }
};
// The share of counter (xcd, sub): how many virtual indices it has (its end) -- groups gg = gi * nxcd + xcd < TG with gi % kSubRanges == sub,
// the recording's last group may be short.
__device__ __forceinline__ void range_share(const ScanArgs& a, WorkRange& w, uint32_t xcd, uint32_t sub)
{
    const uint32_t nxcd = a.nxcd;
    w.nxcd              = nxcd;
    w.xcd               = xcd;
    w.sub               = sub;
    w.range             = xcd * kSubRanges + sub;
    w.counter           = w.range;
    w.first             = 0;
    w.glog              = a.group_log2;
    const uint32_t gsz  = 1u << w.glog;
    const uint32_t tg   = (a.main_chunks + gsz - 1) >> w.glog;
    const uint32_t cx   = tg / nxcd + (xcd < tg % nxcd ? 1u : 0u);                 // groups of this XCD
    const uint32_t ng   = cx > sub ? (cx - sub + kSubRanges - 1) / kSubRanges : 0u; // of this counter
    uint32_t       nv   = ng << w.glog;
    const uint32_t tail = a.main_chunks - ((tg - 1) << w.glog);                    // chunks of the last group (tg >= 1: the launch has work)
    if (ng && tail < gsz && (tg - 1) % nxcd == xcd && ((tg - 1) / nxcd) % kSubRanges == sub) nv -= gsz - tail;
    w.end = nv;
}
__device__ __forceinline__ WorkRange work_range(const ScanArgs& a)
{
    WorkRange      w;
    const uint32_t nxcd = a.nxcd;
    const uint32_t wg   = blockIdx.x / nxcd; // index of this workgroup among those of its XCD
    w.nslot             = gridDim.x / (nxcd * kSubRanges); // the grid is a multiple of nxcd * kSubRanges
    w.slot = wg / kSubRanges;
    range_share(a, w, blockIdx.x % nxcd, wg % kSubRanges);
    return w;
}
// Taking the next work item is two steps for the same reason as Pending: the atomic is issued in front of the prefetch (grab_issue,
// the ticket stays in a vector register of lane 0), its value is read at the end of the trip (grab_value).  (The library is built with
// -amdgpu-atomic-optimizer-strategy=None: the compiler's wave-aggregation of atomics reads the value back on the spot.)
__device__ __forceinline__ uint32_t grab_issue(const ScanArgs& a, const WorkRange& w, int lane)
{
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(&a.work_counters[w.counter * kCounterStride], 1u);
    return v;
}
__device__ __forceinline__ uint32_t grab_value(const WorkRange& w, uint32_t ticket)
{
    return w.first + 2u * w.nslot + (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
}
__device__ __forceinline__ uint32_t grab_chunk(const ScanArgs& a, const WorkRange& w, int lane) { return grab_value(w, grab_issue(a, w, lane)); }

// A work item as the scan loops carry it: the chunk's number in the recording, or kNoChunk.
constexpr uint32_t kNoChunk = 0xFFFFFFFFu;

// The ticket drawn half a trip ago -> the chunk after `next`.  While the wave's counter has work that is its chunk_of(); when the counter
// is dry the wave turns to the pool (below) for the rest of the launch.  (Two forms of work stealing between the XCDs' counters were
// measured and made the kernel slower, 0.211 -> 0.24-0.28 ms: profiles/r04_sweep.txt.)
__device__ __forceinline__ uint32_t take_next(const ScanArgs& a, WorkRange& w, uint32_t ticket, int lane)
{
    // The pool (round 4).  The XCDs get equal shares of the recording but do not run equally fast: the last wave of the fastest XCD left 205 us
    // into a launch, that of the slowest 228 us, and which one is slow changes from launch to launch (profiles/r04_sweep.txt).  So the last
    // sixteenth of a large input belongs to nobody: a wave whose own counter is dry draws from pool counter p = its slot modulo the number of
    // counters -- every pool counter is served by waves of every XCD, those that are done early take more -- chunk main + p + ticket * counters.
    // One returning atomic is waited for at the switch, once per wave; what is lost is the L2 hit on a pool chunk's halo.
    // (The host makes a pool only when every wave is sure to come here: shares of at least four items per wave, make_args in capi.cpp.  As many pool
    // counters as there are own counters, or as a counter has waves if that is fewer: every pool counter has waves that serve it.)
    const uint32_t nranges = w.nxcd * kSubRanges, npool = nranges < w.nslot ? nranges : w.nslot;
    if (w.counter >= kPoolBase)
    {
        const uint32_t c = a.main_chunks + (w.counter - kPoolBase) + (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket) * npool;
        return c < a.total_chunks ? c : kNoChunk;
    }
    const uint32_t v = grab_value(w, ticket);
    if (v < w.end) return w.chunk_of(v);
    if (a.main_chunks < a.total_chunks)
    {
        w.counter        = kPoolBase + w.slot % npool;
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)grab_issue(a, w, lane));
        const uint32_t c = a.main_chunks + (w.counter - kPoolBase) + t * npool;
        return c < a.total_chunks ? c : kNoChunk;
    }
    return kNoChunk;
}

} // namespace
} // namespace adsb_amd
