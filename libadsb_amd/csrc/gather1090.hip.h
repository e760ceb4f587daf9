// gather1090.hip.h -- the ordering pass of the 1090 path as a WAVE-level routine (round 6), shared by the two scan kernels and the stand-alone pass.
//
// What it does is what gather_sorted_kernel did since round 4: a block of kOrderChunks consecutive chunks -> the records of all earlier blocks (the
// sums the scan kernels accumulated), an exclusive prefix over the block's chunk counts, then one lane per record: fetch the raw record, rank it
// among its chunk's records by (offset, pass), apply the repair flip, order the bytes, decode the fields (decode1090.h's function, straight-line
// form), store into the dense arrays.  Record for record the same output.
//
// Why a wave-level form.  Until round 5 a step of the pipelined loop was  scan kernel | 1.9 us | ordering pass 13.6 us | 2.6-3.3 us | next scan:
// 18-19 us per step in which the chip did 3-4 us worth of work (4 354 record waves x ~470 instructions over 1 024 SIMDs).  Now the ordering of
// slot A's scan is done by the waves of the NEXT scan kernel on the stream (slot B's) before they turn to their own chunks: the pass is cut into
// units of a quarter block, a wave takes the units of its index (4 096 waves, 4 096 units per GiB) -- two trips to memory, about 10 us -- and then
// scans.  (First form, measured: a quarter of the waves with a whole block each, drawn from a counter: five dependent trips a block, the pass's
// count reached the host 115 us into the kernel, the host's copy and its next submit came late, and the step went from 0.23 to 0.31 ms.)  Slot A's records were written by an earlier kernel on the same stream, so they are visible to every wave as at any kernel boundary; no
// wave waits for another.  What leaves the kernel WHILE it is still running is the output: the dense arrays go to the host by a copy engine as soon
// as the host sees the pass's count, so
//   * every output store is write-through to memory (sc0 sc1), and a wave waits for its stores' acknowledgements (s_waitcnt vmcnt(0)) before
//   * it adds its unit's record count and a ticket to the slot's state (64-bit atomics, returning, sharded: gather_units); the wave whose ticket is
//     the last -- the pass's LAST FINISHER, whichever unit it had -- stores count | stamp | overflow flag into page-locked host memory (one system-scope
//     store).  Seeing the stamp therefore means every record is in memory.  (Round 5's pass stored it from workgroup nblocks - 1 by index, and the
//     host had to wait for the pass's event on top: ADVICE r05.)
// A scan that has no successor on its stream (the last step of a loop, a live buffer) is ordered by the stand-alone kernel, the same routine with
// one wave per workgroup.
#pragma once

#include "decode1090.h"
#include "scan_common.hip.h"

namespace adsb_amd
{
namespace
{

// ---------------------------------------------------------------------------------------------
// The field decoder of decode1090.h (decode_record) for the ordering pass: the same function of the message, written for a vector unit
// that runs every branch some lane of the wave takes (a wave of 64 records has every kind of message in it, so the byte-wise version
// cost the pass all its branches one after the other: 4.4 of its 13 us).  Straight-line bit-field arithmetic on the message as big-endian
// words B0..B2 (B0 = bytes 0-3, first byte on top), one select per result at the end; the identification's eight characters come out of
// a 64-byte table in LDS (eight byte reads instead of eight chains of comparisons).  Checked against the host build record by
// record (tests: the 4.2 M velocity pairs, every identification character, random records of every DF).
// ---------------------------------------------------------------------------------------------
struct FieldsDev
{
    uint32_t head;     // kind | metype << 8 | mesub << 16 | odd << 24 (adsb_amd_decoded_t's first four bytes)
    uint32_t altitude; // int32
    uint32_t a, b;
};
__device__ __forceinline__ void ais_table_init(uint8_t* tab /* 64, LDS */, uint32_t t)
{
    if (t < 64u) tab[t] = (uint8_t)ais_char(t);
}
__device__ __forceinline__ FieldsDev decode_fields_dev(uint32_t B0, uint32_t B1, uint32_t B2, uint32_t df, const uint8_t* ais /* LDS */)
{
    const uint32_t metype = B1 >> 27, mesub = (B1 >> 24) & 7u;
    // DF0/4/20: the 13-bit AC field, bytes 2-3 (:440-466)
    const uint32_t ac13 = B0 & 0x1FFFu;
    const int      n13  = (int)(((ac13 & 0x1F80u) >> 2) | ((ac13 & 0x20u) >> 1) | (ac13 & 0xFu));
    const int      alt13 = ((ac13 & 0x40u) || !(ac13 & 0x10u)) ? 0 : n13 * 25 - 1000;
    // airborne position (:622-630, 470-486): AC12 = bytes 5 and 6's high nibble, F flag, 17-bit raw latitude and longitude
    const uint32_t ac12  = (B1 >> 12) & 0xFFFu;
    const int      alt12 = (ac12 & 0x10u) ? (int)(((ac12 >> 5) << 4) | (ac12 & 0xFu)) * 25 - 1000 : 0;
    const uint32_t lat   = ((B1 & 0x3FFu) << 7) | (B2 >> 25), lon = (B2 >> 8) & 0x1FFFFu, odd = (B1 >> 10) & 1u;
    // airborne velocity (:631-660)
    const int ew = (int)((B1 >> 8) & 0x3FFu), ns = (int)(((B1 & 0x7Fu) << 3) | (B2 >> 29));
    const int n  = ns * ns + ew * ew;
    int       v  = (int)__builtin_sqrtf((float)n); // within one of the integer square root (n < 2^21 is exact in a float)
    v -= (v * v > n) ? 1 : 0;
    v += ((v + 1) * (v + 1) <= n) ? 1 : 0;
    const int h = v ? heading_of((B1 & (1u << 18)) ? -ew : ew, (B1 & 0x80u) ? -ns : ns) : 0;
    // identification (:608-619): eight 6-bit characters in bytes 5-10
    const uint32_t c03 = B1 & 0xFFFFFFu, c47 = B2 >> 8;
    const uint32_t ia = (uint32_t)ais[c03 >> 18] | ((uint32_t)ais[(c03 >> 12) & 63u] << 8) | ((uint32_t)ais[(c03 >> 6) & 63u] << 16) | ((uint32_t)ais[c03 & 63u] << 24);
    const uint32_t ib = (uint32_t)ais[c47 >> 18] | ((uint32_t)ais[(c47 >> 12) & 63u] << 8) | ((uint32_t)ais[(c47 >> 6) & 63u] << 16) | ((uint32_t)ais[c47 & 63u] << 24);

    const bool is_alt = df == 0u || df == 4u || df == 20u, es = df == 17u;
    const bool is_id = es && metype - 1u < 4u, is_pos = es && metype - 9u < 10u, is_vel = es && metype == 19u && mesub - 1u < 2u;
    FieldsDev  f;
    const uint32_t kind = is_alt ? (uint32_t)ADSB_AMD_K_ALTITUDE : is_id ? (uint32_t)ADSB_AMD_K_IDENT : is_pos ? (uint32_t)ADSB_AMD_K_POSITION : is_vel ? (uint32_t)ADSB_AMD_K_VELOCITY : 0u;
    f.head     = kind | (metype << 8) | (mesub << 16) | ((is_pos ? odd : 0u) << 24);
    f.altitude = (uint32_t)(is_alt ? alt13 : is_pos ? alt12 : 0);
    f.a        = is_id ? ia : is_pos ? lat : is_vel ? (uint32_t)v : 0u;
    f.b        = is_id ? ib : is_pos ? lon : is_vel ? (uint32_t)h : 0u;
    return f;
}

// LDS a gathering wave needs: the chunks' first records (cstart), the identification characters.  In the scan kernels it lies in the image
// area, which the wave has not started to use.
struct alignas(16) GatherLds
{
    uint32_t cstart[kOrderChunks + 1];
    uint8_t  ais[64];
};

// sixteen bytes to memory, written through every cache on the way (system scope): the copy engine that fetches them does not look into an L2.
// (The s_nop: a store of more than eight bytes reads its data registers over the cycles AFTER it issues, and a vector instruction must not write them
// in the two cycles behind it -- gfx940-class hazard; the compiler keeps that distance for its own stores and cannot see into this one.  Without
// it the lanes of a wave's last quarter left with the NEXT value of a register: a record's offset word held the first word of its second half.)
__device__ __forceinline__ void store16_through(void* p, uint4 v)
{
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}

// The pass is cut into UNITS: `split` per block of kOrderChunks chunks, unit (B, q) moving the block's records r with (r / 64) % split == q.
// Every wave that takes part has its units by index (u = wave, wave + waves, ...: no counter to draw from) and works out its block's prefix for
// itself.  Inside a scan kernel a unit is a whole block (split 1: five dependent trips of 64 records for a quiet band, ~40 us for the one wave in
// four that does it, beside three that scan); the stand-alone pass, where nothing else runs, cuts a block into four.

// One unit of the ordering pass, by one wave.  Uniform control flow; every lane must call.  Returns the records the unit moved.
__device__ __forceinline__ uint32_t gather_unit(const GatherArgs& ga, uint32_t B, uint32_t q, uint32_t kGatherSplit, GatherLds& lds, uint32_t lane)
{
    // records in earlier blocks: the block sums the scan kernels accumulated, one on each 128-byte line (a second level of sums, an entry per sixteen
    // blocks, was built and measured: every wave of a scan works in the same super-block at any time, 131 072 more atomics queued on one line after
    // another, and the scan kernel went from 0.205 to 0.31 ms)
    uint32_t before = 0;
    for (uint32_t b = lane; b < B; b += 64u) before += ga.block_sums[b * kSumStride];
    // this block's chunk counts, four chunks a lane
    const uint32_t c0 = B * kOrderChunks + 4u * lane;
    uint32_t       n[4];
    if (c0 + 4u <= ga.nchunks)
    {
        const uint4 v = *reinterpret_cast<const uint4*>(ga.chunk_dir + c0); // (the directory is 16-byte aligned and c0 a multiple of four)
        n[0] = v.x, n[1] = v.y, n[2] = v.z, n[3] = v.w;
    }
    else
    {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) n[k] = c0 + k < ga.nchunks ? ga.chunk_dir[c0 + k] : 0u;
    }
    const uint32_t base = wave_sum(before);
    const uint32_t mine = n[0] + n[1] + n[2] + n[3];
    const uint32_t incl = wave_incl_scan_add(mine);
    const uint32_t tot  = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    {
        const uint32_t at = incl - mine;
        *reinterpret_cast<uint4*>(&lds.cstart[4u * lane]) = make_uint4(at, at + n[0], at + n[0] + n[1], at + n[0] + n[1] + n[2]);
        if (lane == 0) lds.cstart[kOrderChunks] = tot;
        ais_table_init(lds.ais, lane);
    }
    wave_lds_fence();

    // A trip: this lane's record of 64 consecutive ones and the keys of its chunk's first four records.  The loads of the NEXT trip are issued before
    // the current one is worked on (one trip ahead: sixteen registers), so that a block is not five times "records in, records out" one after the other.
    struct Trip
    {
        bool         valid;
        uint32_t     first, nn, idx, ch;
        const uint4* src;
        uint4        lo, hi;
        uint2        key4[4];
    };
    auto fetch = [&](uint32_t trip0) -> Trip
    {
        Trip           t;
        const uint32_t r = trip0 + lane; // this lane's record among the block's
        t.valid          = r < tot;
        // its chunk: the last one whose first record is not behind r (chunks without records share their successor's first record)
        uint32_t cc = 0;
#pragma unroll
        for (uint32_t step = kOrderChunks / 2; step >= 1; step >>= 1)
            if (lds.cstart[cc + step] <= (t.valid ? r : 0u)) cc += step;
        t.first = lds.cstart[cc], t.nn = lds.cstart[cc + 1] - t.first, t.idx = t.valid ? r - t.first : 0u;
        t.ch    = B * kOrderChunks + cc;
        t.src   = reinterpret_cast<const uint4*>(ga.chunk_records + (uint64_t)t.ch * ga.cap);
        t.lo = t.hi = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) t.key4[j] = make_uint2(0, 0);
        if (t.valid)
        {
            t.lo = t.src[2 * t.idx], t.hi = t.src[2 * t.idx + 1];
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) t.key4[j] = *reinterpret_cast<const uint2*>(&t.src[2 * (j < t.nn ? j : t.idx)]);
        }
        return t;
    };
    uint32_t       moved  = 0;
    const uint32_t stride = kGatherSplit * 64u;
    Trip           cur    = fetch(q * 64u);
    for (uint32_t trip0 = q * 64u; trip0 < tot; trip0 += stride)
    {
        Trip nxt = cur;
        if (trip0 + stride < tot) nxt = fetch(trip0 + stride); // (uniform)
        if (cur.valid)
        {
            // rank among the chunk's records by (offset, pass): their keys four at a time (a chunk seldom has more)
            const uint32_t key  = (cur.lo.x << 1) | ((cur.lo.y >> 16) & 1u);
            uint32_t       rank = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) rank += (((cur.key4[j].x << 1) | ((cur.key4[j].y >> 16) & 1u)) < key) ? 1u : 0u;
            for (uint32_t k0 = 4; k0 < cur.nn; k0 += 4)
            {
                uint2 w[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) w[j] = *reinterpret_cast<const uint2*>(&cur.src[2 * (k0 + j < cur.nn ? k0 + j : cur.idx)]);
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) rank += (((w[j].x << 1) | ((w[j].y >> 16) & 1u)) < key) ? 1u : 0u;
            }
            const uint32_t buffer = cur.ch / ga.chunks_per_buf;
            const size_t   out    = (size_t)base + cur.first + rank;
            // raw -> adsb_amd_record_t: apply the 1-bit repair, order the bytes, pull the address out
            const uint32_t df = cur.lo.y & 0xFFu, nbits = (cur.lo.y >> 8) & 0xFFu, flags = (cur.lo.y >> 16) & 0xFFu;
            const int      errorbit = (int)(cur.lo.y >> 24) - 1;
            uint4          h = cur.hi;
            if (errorbit >= 0)
            {
                const uint32_t m = 1u << (errorbit & 31);
                if (errorbit < 32) h.x ^= m;
                else if (errorbit < 64) h.y ^= m;
                else if (errorbit < 96) h.z ^= m;
                else h.w ^= m;
            }
            // the message as big-endian words (first bit on top) and as bytes in memory order
            const uint32_t B0 = __builtin_bitreverse32(h.x), B1 = __builtin_bitreverse32(h.y), B2 = __builtin_bitreverse32(h.z), B3 = __builtin_bitreverse32(h.w);
            const uint32_t m0 = __builtin_bswap32(B0), m1 = __builtin_bswap32(B1), m2 = __builtin_bswap32(B2), m3 = __builtin_bswap32(B3); // bytes 0-3, 4-7, 8-11, 12-13
            const uint32_t addr = (flags & ADSB_AMD_F_NEEDS_ICAO) ? cur.lo.z : (B0 & 0xFFFFFFu);
            uint4 o0;
            o0.x = buffer;
            o0.y = cur.lo.x;
            o0.z = addr;
            o0.w = (cur.lo.w & 0xFFFFu) | (nbits << 16) | (((uint32_t)errorbit & 0xFFu) << 24);
            if (ga.dense)
            {
                uint4* o = reinterpret_cast<uint4*>(ga.dense + out);
                store16_through(o, o0);
                store16_through(o + 1, make_uint4(df | (flags << 8) | (m0 << 16), (m0 >> 16) | (m1 << 16), (m1 >> 16) | (m2 << 16), (m2 >> 16) | (m3 << 16)));
            }
            // the stateless half of DecodeModesMessage, so that the host's sequential pass decodes nothing
            const FieldsDev d = decode_fields_dev(B0, B1, B2, df, lds.ais);
            if (ga.decoded) store16_through(ga.decoded + out, make_uint4(d.head, d.altitude, d.a, d.b));
            if (ga.packed)
            { // the record's first sixteen bytes, then df, flags, kind, odd and the decoded values (adsb_amd_packed_t)
                uint4* o = reinterpret_cast<uint4*>(ga.packed + out);
                store16_through(o, o0);
                store16_through(o + 1, make_uint4(df | (flags << 8) | ((d.head & 0xFFu) << 16) | (d.head & 0xFF000000u), d.altitude, d.a, d.b));
            }
            moved++;
        }
        cur = nxt;
    }
    wave_lds_fence(); // (the next unit's prefix overwrites cstart)
    return wave_sum(moved);
}

// A unit and what follows it: a share of the housekeeping for the slot's next scan, the wait for the stores' acknowledgements, the tickets.
// State words (GatherArgs::state, zeroed by the slot's scan kernel): the master word at [0..1] -- records | shards finished << 32 --, the overflow
// flag at [2], and kGatherShards shard words, each on a line of its own, at
// [kStateShard0 + 32 s ..] -- records | units finished << 32.  A unit's ticket goes to shard u % kGatherShards (4 096 tickets on ONE word would
// queue for 30 us behind each other); the wave that finishes a shard adds the shard's records and a ticket to the master word; the wave that
// finishes the last shard is the pass's LAST FINISHER and tells the host.
__device__ __forceinline__ void gather_unit_and_ticket(const GatherArgs& ga, uint32_t u, uint32_t units, uint32_t split, GatherLds& lds, uint32_t lane)
{
    // housekeeping, spread over the units: the slot's other sum array (all of it: the next input may be larger) and its work counters start from zero
    for (uint32_t k = u * 64u + lane; k < 2u * ga.next_entries; k += units * 64u) ga.next_block_sums[(k >> 1) * kSumStride + (k & 1u)] = 0;
    if (u == 0)
        for (uint32_t k = lane; k < kWorkCounters; k += 64u) ga.work_counters[k * kCounterStride] = 0;
    const uint32_t B = u / split, q = u % split;
    const uint32_t moved = gather_unit(ga, B, q, split, lds, lane);
    const bool     over  = q == 0 && ga.block_sums[B * kSumStride + 1] != 0; // some chunk of this block overflowed its region
    // Everything this wave stored has been acknowledged -- the records by memory, the zeroes by the L2 -- before its ticket counts.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t last_total = 0xFFFFFFFFu; // lane 0 of the pass's last finisher: the pass's record count
    bool     any_over   = false;
    if (lane == 0)
    {
        if (over)
        { // (a returning atomic whose value is waited for: performed before the ticket below is drawn)
            const uint32_t was = atomicOr(&ga.state[2], 1u);
            asm volatile("" ::"v"(was));
        }
        const uint32_t            shard   = u % kGatherShards, nshards = units < kGatherShards ? units : kGatherShards;
        const uint32_t            in_shard = (units - shard + kGatherShards - 1u) / kGatherShards; // units u' < units with u' % kGatherShards == shard
        unsigned long long* const sw = reinterpret_cast<unsigned long long*>(ga.state + kStateShard0 + 32u * shard);
        const unsigned long long  was = atomicAdd(sw, (1ull << 32) | (unsigned long long)moved);
        if ((uint32_t)(was >> 32) == in_shard - 1u)
        {
            const unsigned long long m = atomicAdd(reinterpret_cast<unsigned long long*>(ga.state), (1ull << 32) | (unsigned long long)((uint32_t)was + moved));
            if ((uint32_t)(m >> 32) == nshards - 1u)
            {
                last_total = (uint32_t)m + (uint32_t)was + moved;
                any_over   = atomicOr(&ga.state[2], 0u) != 0u;
            }
        }
    }
    if (last_total != 0xFFFFFFFFu && ga.host_word)
        __hip_atomic_store(ga.host_word, (unsigned long long)last_total | ((unsigned long long)(((ga.stamp & 0x7FFFFFFFu) << 1) | (any_over ? 1u : 0u)) << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The units of a pass this wave takes by index: `wave` of the `waves` that take part (no counter to draw from).
__device__ __forceinline__ void gather_units(const GatherArgs& ga, uint32_t wave, uint32_t waves, uint32_t split, GatherLds& lds, uint32_t lane)
{
    const uint32_t units = ga.nblocks * split;
    if (wave >= units) return;
    __builtin_amdgcn_s_setprio(2); // short dependent chains beside scanning waves' dense phases (as the scan kernels' own sparse phases)
    for (uint32_t u = wave; u < units; u += waves) gather_unit_and_ticket(ga, u, units, split, lds, lane);
    __builtin_amdgcn_s_setprio(0);
}

// In front of a scan kernel's own work (single-wave workgroups, blockIdx.x = the wave): the ordering pass of the other slot's scan, by about as many
// waves as it has blocks, a block each -- spread over the XCDs and, as far as the order in which workgroups are placed allows to say, over the CUs: of
// the rounds in which every CU of the chip receives a workgroup, every spacing-th, so that a SIMD has one gathering wave beside three scanning ones.
// (Every wave gathering a quarter block first, the form before this one, cost the kernel 11 us -- as much as the separate pass had: 4 096 waves
// waiting out the same three memory round trips with nothing else on the chip.  And the pass BEHIND the scan -- the waves that have run out of chunks
// drawing quarter blocks from a counter, to fill the 20-30 us in which a launch's waves end one after the other -- made the kernel 55 us longer:
// cut that fine the pass is 100 wave-milliseconds of waiting for memory, more than that window holds, and a unit drawn late ends 25 us after the last
// chunk.  profiles/r06_ordering_in_kernel.txt)
__device__ __forceinline__ void gather_in_front(const GatherArgs& ga, uint32_t ncu, GatherLds& lds, uint32_t lane)
{
    if (ga.nblocks == 0) return;
    const uint32_t per_round = ncu ? ncu : 1u, rounds = gridDim.x / per_round; // whole rounds
    if (rounds == 0)
    { // fewer waves than CUs: all of them
        gather_units(ga, blockIdx.x, gridDim.x, 1u, lds, lane);
        return;
    }
    const uint32_t want    = (ga.nblocks + per_round - 1u) / per_round;
    const uint32_t spacing = want >= rounds ? 1u : rounds / want;
    const uint32_t r       = blockIdx.x / per_round;
    if (r >= rounds || r % spacing != 0u) return;
    gather_units(ga, (r / spacing) * per_round + blockIdx.x % per_round, ((rounds + spacing - 1u) / spacing) * per_round, 1u, lds, lane);
}

// the state words of a slot start from zero (by the first workgroup of the slot's scan kernel; its pass runs in a later kernel)
__device__ __forceinline__ void gather_state_zero(uint32_t* state, uint32_t lane)
{
    if (lane < kGatherShards) state[kStateShard0 + 32u * lane] = 0, state[kStateShard0 + 32u * lane + 1u] = 0;
    if (lane < 4u) state[lane] = 0;
}

} // namespace
} // namespace adsb_amd
