// scan1090.h -- internal launch interface between the C-ABI layer (capi.cpp) and the gfx950 kernels (scan1090.hip).
#pragma once

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "adsb_amd.h"

namespace adsb_amd
{

// Geometry of one work item ("chunk"): one wavefront demodulates kChunk consecutive preamble positions
// of one reference buffer out of an LDS-staged window of kChunk + kHalo samples.
constexpr int kLanes      = 64;
constexpr int kRowSamples = 512; // one 16-byte load per lane = 8 IQ samples per lane = 512 per wavefront
constexpr int kRows       = 8;   // 1 KiB load rows per chunk (6 rows = 3072 positions, 20 waves a CU: measured, not kept, profiles/r04_sweep.txt)
constexpr int kWavesPerCu = 16;  // single-wave workgroups of the persistent scan kernels a CU holds (LDS-limited)
constexpr int kChunk      = kRows * kRowSamples; // 4096 positions
constexpr int kHalo       = 256;                 // >= 240 samples read past the last position, half a row
constexpr int kFrameSpan  = 240;                 // (8 + 112) * 2 samples: reference loop bound (ADSB1090.cpp:772)

// LDS image of a chunk's s = (I-127)^2 + (Q-127)^2 values: the two halves of the chunk are interleaved, dword q holds
// (s[q] | s[q + 2048] << 16) for q = 0 .. 2047.  A packed 16-bit operation on dword q therefore works on two preamble
// positions 2048 samples apart, and the operand "sample q + a" of BOTH positions is simply dword q + a, whatever the parity
// of a: no funnel shifts, and every sliding intermediate (max(s[q], s[q+2]), ...) is computed once per dword and used at every
// offset.  The image continues for kHalo more dwords, (s[2048 + m] | s[4096 + m] << 16): the low halves repeat the start of the
// upper half, the high halves are the halo behind the chunk, so the 240-sample window of ANY position is one linear run of
// halves: sample p of the chunk (0 <= p < kChunk + kHalo) lives at uint16_t index tile_index(p) = 2 (p % 2048) + p / 2048
// for p < 4096 and 2 (p - 2048) + 1 beyond, i.e. consecutive samples of a window are always 2 halves apart.
constexpr int kHalfChunk   = kChunk / 2;                    // 2048
constexpr int kImageDwords = kHalfChunk + kHalo;            // 2304
constexpr int kFrontSlot16 = 2 * kImageDwords;              // 4608: the sample just before the chunk (g0 - 1)
constexpr int kTileDwords  = kImageDwords + 4;              // 2308 dwords = 9232 bytes

struct ScanArgs
{
    const uint8_t*     iq;           // device, 16-byte aligned
    uint64_t           buf_stride;   // bytes between consecutive reference buffers
    uint32_t           buf_samples;  // N: samples per reference buffer
    uint32_t           nbuf;
    uint32_t           chunks_per_buf;
    uint32_t           group_log2;   // log2 of the chunks dealt out together (scan_common.hip.h, WorkRange): 4 for inputs that fill the chip, 0 for small ones
    uint32_t           cpb_magic;    // floor(2^32 / chunks_per_buf) (0xFFFFFFFF for one chunk per buffer): chunk -> (buffer, chunk in buffer) without a division
    uint32_t           total_chunks;
    uint32_t           main_chunks;  // chunks [0, main_chunks) are dealt out by the XCDs' own counters (scan_common.hip.h, WorkRange), the rest -- the pool,
                                     // about a sixteenth, none for small inputs -- by counters that waves of every XCD draw from once their own is dry
    const uint32_t*    crc_tab;      // 112 entries (ModesChecksumTable semantics)
    adsb_amd_record_t* chunk_records; // total_chunks * cap raw records: one region of `cap` per chunk.  (Per-wave logs instead of per-chunk regions,
                                      // so that the ordering pass reads dense memory, were measured in round 4: the scan 3 % slower, the pass no
                                      // faster once it had a lane per record -- profiles/r04_sweep.txt; not kept.)
    uint32_t*          chunk_dir;     // one word per chunk: records kept (<= cap)
    uint32_t           cap;           // records per chunk region
    uint32_t*          work_counters; // kSubRanges per XCD, counter c at [32 * c] (own cache line each), zero when the scan starts; the ordering pass zeroes them again
    uint32_t           nxcd;          // XCDs of the device (hipDeviceAttributeNumberOfXccs; 8 on MI355X), <= kMaxXcd
    uint32_t           ncu;           // compute units of the device (256 on MI355X)
    unsigned long long* stamps;       // NULL in the product build; measurement builds (diag.hip.h, tools/stamps.py): {scan wave in, scan wave out,
                                      // ordering pass in, out} of this launch per wave, 100 MHz clock
    uint32_t*          block_sums;    // one padded entry (kSumStride words) per kOrderChunks chunks, zero when the scan starts: [0] += records of a
                                      // finished chunk (clamped to cap), [1] |= 1 when a chunk found more than cap
};
constexpr uint32_t kOrderChunks = 256; // chunks per entry of block_sums = per block of the ordering pass
constexpr uint32_t kSumStride   = 32; // words between entries: an entry per 128-byte line (one atomic per chunk lands on it)
inline size_t sums_entries(size_t chunks) { return (chunks + kOrderChunks - 1) / kOrderChunks; } // entries of one sum array for `chunks` chunks
// a slot's ordering state (GatherArgs::state; gather1090.hip.h): four words, then kGatherShards shard words 32 words apart
constexpr uint32_t kGatherShards = 32, kStateShard0 = 32, kStateWords = kStateShard0 + 32 * kGatherShards;

constexpr uint32_t kSubRanges   = 4;  // work counters per XCD
constexpr uint32_t kMaxXcd      = 16;
constexpr uint32_t kWorkCounters = 2 * kSubRanges * kMaxXcd; // the XCDs' own, then as many for the pool (from kPoolBase)
constexpr uint32_t kPoolBase     = kSubRanges * kMaxXcd;
constexpr uint32_t kCounterStride = 32; // words between work counters (a 128-byte line each)

inline uint32_t chunks_per_buffer(uint32_t buf_samples)
{
    if (buf_samples <= (uint32_t)kFrameSpan) return 0;
    uint32_t positions = buf_samples - (uint32_t)kFrameSpan;
    return (positions + (uint32_t)kChunk - 1) / (uint32_t)kChunk;
}

// Waves of the persistent scan kernels for this input (both rates): as many single-wave workgroups as the LDS lets the chip hold (16 per
// CU), in whole (XCD, sub-range) units; fewer for inputs with fewer chunks.
inline uint32_t scan_grid(const ScanArgs& a)
{
    const uint32_t unit = a.nxcd * kSubRanges; // every (XCD, sub-range) gets the same number of waves
    uint32_t       grid = (a.ncu * (uint32_t)kWavesPerCu / unit) * unit;
    if (grid == 0) grid = unit;
    if (grid > a.total_chunks) grid = ((a.total_chunks + unit - 1u) / unit) * unit;
    return grid;
}

// The ordering pass of one scan (gather1090.hip.h): the scan's raw records, chunk directory and block sums -> the dense sorted arrays, with the
// field decode.  Run by the next scan kernel on the stream (launch_scan1090 / launch_scan2400, `attached`) or on its own (launch_gather1090).
struct GatherArgs
{
    const adsb_amd_record_t* chunk_records = nullptr; // ScanArgs::chunk_records, chunk_dir, block_sums, total_chunks, cap, chunks_per_buf of the scan to order
    const uint32_t*          chunk_dir     = nullptr;
    const uint32_t*          block_sums    = nullptr;
    uint32_t                 nchunks       = 0;
    uint32_t                 nblocks       = 0; // ceil(nchunks / kOrderChunks); 0: nothing to order
    uint32_t                 cap           = 0;
    uint32_t                 chunks_per_buf = 1;
    adsb_amd_record_t*       dense         = nullptr; // the arrays to produce (any may be NULL)
    adsb_amd_decoded_t*      decoded       = nullptr;
    adsb_amd_packed_t*       packed        = nullptr;
    uint32_t*                state         = nullptr; // the slot's kStateWords words, zero when the pass starts (its scan kernel zeroed them): see gather_units
    uint32_t*                next_block_sums = nullptr; // the slot's other sum array (`next_entries` padded entries), zeroed for the slot's next scan
    uint32_t                 next_entries  = 0;
    uint32_t*                work_counters = nullptr; // the slot's work counters, zeroed for its next scan
    unsigned long long*      host_word     = nullptr; // device address of 8 bytes of page-locked host memory (may be NULL): the pass's last finisher stores
                                                      // count | (stamp << 1 | overflow flag) << 32 there, one system-scope store, after every record is in memory
    uint32_t                 stamp         = 0;       // 31 bits
    unsigned long long*      stamps        = nullptr; // measurement builds (ScanArgs::stamps)
};

// Demodulation kernel (fills the raw record regions and the chunk directory; zeroes `total_and_overflow`, the slot's GatherArgs::state, kStateWords words).
// `start` / `stop` (either may be NULL): events that take the kernel's own start and end times -- they ride on the dispatch
// (hipExtLaunchKernelGGL), where two hipEventRecord calls around the launch are packets of their own on the stream, 3-5 us each.
// `attached` (may be NULL): the ordering pass of an EARLIER scan on the same stream, done by this kernel's waves before their own work.
hipError_t launch_scan1090(const ScanArgs& a, uint32_t* total_and_overflow, hipStream_t stream, hipEvent_t start = nullptr, hipEvent_t stop = nullptr,
                           const GatherArgs* attached = nullptr);
// The ordering pass as a launch of its own.  `done` (may be NULL): event that takes the pass's end (riding on its dispatch like the scan's two).
hipError_t launch_gather1090(const GatherArgs& ga, hipStream_t stream, hipEvent_t done = nullptr);
// the field decoder of the ordering pass over an arbitrary device record array (parity helper)
hipError_t launch_decode1090(const adsb_amd_record_t* rec, adsb_amd_decoded_t* out, size_t n, hipStream_t stream);

// magnitudes exactly as the reference computes them (parity helper)
hipError_t launch_magnitude1090(const uint8_t* iq, uint16_t* mag, size_t nsamples, hipStream_t stream);

// UAT978 phase LUT map (UAT978.cpp:52): phi[k] = lut[I | Q<<8]
hipError_t launch_phase978(const uint8_t* iq, uint16_t* phi, size_t nsamples, const uint16_t* lut65536, hipStream_t stream);

// ---- the 2.4 MS/s mode (scan2400.hip; definition: oracle/oracle2400.c).  Same ScanArgs, same raw records, same ordering pass.
uint32_t   chunks_per_buffer_2400(uint32_t buf_samples);
hipError_t launch_scan2400(const ScanArgs& a, uint32_t* total_and_overflow, hipStream_t stream, hipEvent_t start = nullptr, hipEvent_t stop = nullptr,
                           const GatherArgs* attached = nullptr);

// host-side table builders (exact integer / polynomial arithmetic, no reference text)
void build_crc_table(uint32_t* tab /* 112 */);

} // namespace adsb_amd
