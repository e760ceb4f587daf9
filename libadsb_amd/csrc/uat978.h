// uat978.h -- device-side records and launch prototypes of the UAT 978 path (see uat978.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adsb_amd
{
constexpr int kUatSyncBits    = 36;
constexpr int kUatCheckBits   = 18;
constexpr int kUatLongBytes   = 48;
constexpr int kUatShortBytes  = 30;
constexpr int kUatUplinkBytes = 552;
constexpr int kUatUplinkBits  = kUatUplinkBytes * 8;
constexpr uint32_t kUatDemodRanges = 64;

// one per 18-bit match, in candidate order; variant v = frame taken from sample index + v.  16 bytes: what the host reads for a match whose frame the
// scan loop takes (round 6: the two register windows, which only the host's own scan loop -- the fall-back -- reads, moved to uat_win_t, a parallel
// array that goes to the host only when that loop runs: 1.25 MB per GiB less on the host link per call).  The corrected ADS-B frame bytes live in a
// second parallel array (kUatPayloadStride bytes per match), read only when a frame is handed to a listener.
struct uat_rec_t
{
    uint32_t index;       // sample index of the first sync bit
    int16_t  skip;        // bits the scan loop jumps when it takes this frame: 276 short, 420 long, 4452 uplink; 0 = no frame
    uint8_t  rs;          // corrected symbols of the frame taken (uplink: sum over the six blocks); 255 = no frame
    uint8_t  kind;        // 0 = ADS-B sync word, 1 = uplink sync word
    uint8_t  variant;     // the frame was taken from sample index + variant (the reference demodulates both and keeps the one
                          // with fewer corrections, the first on a tie); 2 = neither decodes
    uint8_t  pad0[3];
    uint32_t slot;        // uplink: 432-byte slot of the decoded payload in the side array
};
static_assert(sizeof(uat_rec_t) == 16, "record layout");
struct uat_win_t
{
    uint64_t window;      // sign bits from sample 2 * (index >> 1) on, low word = even samples (register 0), high word = odd
                          // samples (register 1), bit k = k-th bit time: both 18-bit registers at detection time
    uint64_t after;       // the same from bit (index >> 1) + skip + 1 on: what enters the registers after the jump
};
static_assert(sizeof(uat_win_t) == 16, "window layout");
constexpr uint32_t kUatPayloadStride = 40; // ADS-B: corrected frame bytes 0..33 (18 of them for a short frame: skip == 276), 
// The matches of a stream, binned by position while they are found (round 6): bin = sample index >> kUatBinShift, kUatBinCap slots each.
constexpr uint32_t kUatBinShift = 15, kUatBinCap = 16;

// A frame the scan loop reaches only through stale register bits in the 17 bits after a jump: its position need not be a match of
// the stream, so it is not in the match list.  The wave that demodulated the frame before it (`parent`, a position in the ordered
// list) demodulates it as well; `seq` orders the extras of one parent.  rec.variant == 2: the position did not decode, the loop
// went on to the next step.
struct uat_extra_t
{
    uat_rec_t rec;
    uint32_t  parent, seq, pad[2];
};
static_assert(sizeof(uat_extra_t) == 32, "extra record layout");
constexpr uint32_t kUatExtraCap = 4096; // per call (about 19 per GiB of frame-dense stream); past it the host walks the loop itself
constexpr uint32_t kUatEnd      = 0xFFFFFFFFu; // successor / emit value: none
// counts[]: what the kernels of one call report
enum UatCount : uint32_t
{
    kUatCountMatches = 0,  // 18-bit matches found
    kUatCountUplinkSlots,  // 432-byte payload slots handed out
    kUatCountUplinkMatches,// matches on the uplink check word (ordering pass)
    kUatCountExtras,       // uat_extra_t entries reserved
    kUatCountOverflow,     // != 0: an extra or its payload slot did not fit; the decisions below are not valid
    kUatCountFinalBit,     // the largest `next bit` over the frames the loop takes
    kUatCountTaken,        // frames of the match list the loop takes
    kUatCountBinOverflow,  // != 0: some bin of the match search held more than kUatBinCap matches: the bins are not the list, order the list itself
    kUatCountWords = 16
};

struct RsTables;

struct UatArgs
{
    const uint16_t* in;  // u8 IQ pairs as u16 (phases_given == 0) or u16 phases (phases_given == 1)
    const uint16_t* lut; // 65536-entry phase LUT
    const RsTables* rs_tables;
    uint64_t        nsamples;
    uint32_t        ncu; // compute units of the device (read when the handle is made; 0: 256)
    int             phases_given;
    uint64_t*       signs; // phases path only: ceil(nsamples / 64) + 2 words
    uint32_t*       cand;
    uint32_t        cand_cap;
    uint32_t*       counts; // kUatCountWords words, see UatCount
    uint32_t*       up_list; // cand_cap entries: positions of the uplink matches in the ordered list
    uat_rec_t*      recs;   // cand_cap entries
    uat_win_t*      wins;   // parallel to recs: the register windows (for the host's own scan loop)
    uint8_t*        payloads; // cand_cap x kUatPayloadStride bytes, parallel to recs
    uint8_t*        uplink_payloads; // uplink_cap x 432 bytes
    uint32_t        uplink_cap;
    uint32_t*       demod_work; // kUatDemodRanges work counters, 32 words apart
    uint32_t        single_word; // launch_uat978_demod with cand == nullptr and ncand == 1: the one match word to demodulate
    // the scan loop's decisions on the device (launch_uat978_decide); all cand_cap entries, positions are those of the ordered list
    int64_t         lenbits;   // bits the loop examines: nsamples / 2 - (36 + 4416)
    // A part of a longer stream (adsb_amd_uat_part_*): the loop is clean at `entry` when it comes into this part, so the first start bit
    // it can fire at is first_bit = max(entry - 17, the part's first own bit) (a whole stream: 1 -- a match whose 18 bits end at bit 17
    // is never looked at); start bits from end_bit on belong to the next part (a whole stream: none, the lenbits rule ends the walk)
    int64_t         first_bit, end_bit;
    uint32_t*       next_bit;  // per match: the bit the loop examines next with clean registers once it has taken this frame (and any
                               // frames the stale registers fired on behind it); 0 = no frame here
    uat_extra_t*    extras;    // kUatExtraCap entries
    uint8_t*        extra_payloads; // kUatExtraCap x kUatPayloadStride
    uint32_t        extra_cap;  // entries of the two that may be used (<= kUatExtraCap; tests lower it to reach the overflow path)
    uint32_t*       succ;      // per start bit (its first match): the match the loop reaches next, kUatEnd = none
    uint32_t*       exit_of;   // the first match outside the node's block of kUatDecideNodes on that path
    uint32_t*       emit_of;   // the match whose frame the loop takes at this start bit, kUatEnd = none
    uint32_t*       marks;     // bit per match: the loop takes its frame
    // the bins of the match search (IQ input): fill counts of this call, the array the ordering pass zeroes for the next call (the two swap
    // roles per call), how many words each of the two holds, and kUatBinCap slots per bin
    uint32_t*       bin_fill;
    uint32_t*       bin_fill_next;
    uint32_t        bins_cap;
    uint32_t*       bin_slots;
};
constexpr uint32_t kUatDecideNodes = 4096;

hipError_t launch_uat978(const UatArgs& a, hipStream_t stream);                       // signs + 18-bit match
hipError_t launch_uat978_order(const UatArgs& a, uint32_t ncand, uint32_t* scratch, uint32_t* sorted, hipStream_t stream);
// the same from the bins the match search filled (a.bin_fill / a.bin_slots; no bin ran over): one launch
hipError_t launch_uat978_order_bins(const UatArgs& a, uint32_t ncand, uint32_t* sorted, hipStream_t stream);
hipError_t launch_uat978_demod(const UatArgs& a, uint32_t ncand, bool ordered, hipStream_t stream); // one wave per candidate
// which frames the loop takes.  wide: workgroups of 1024 threads (a call that has the chip to itself) instead of 256 (calls in flight: four waves
// find room beside another call's demodulation pass, sixteen wait for it)
hipError_t launch_uat978_decide(const UatArgs& a, uint32_t ncand, const uint32_t* sorted, hipStream_t stream, bool wide);
hipError_t launch_uat978_rs_selftest(const RsTables* tables, int kind, uint8_t* words, int* results, int count, hipStream_t stream);
} // namespace adsb_amd
