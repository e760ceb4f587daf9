// resolver1090.hpp -- host half of the 1090 path: candidate records -> accepted frames -> aircraft state.
//
// The GPU emits every (offset, pass) the reference *could* accept; what it *does* accept depends on state that is
// inherently sequential (reference ADSB1090.cpp:886-957): a frame accepted at j hides the next 128/240 offsets,
// AP-type DFs are valid only if their address was recently seen in a clean DF11/17 (:195-207, :396-435), and the
// retry slice is only looked at when the first slice was not accepted.  Resolver1090 applies exactly those rules in three
// passes over batches of accepted frames:
//   1. sequential: skip-ahead, ICAO gating, and which even/odd pair a position frame completes (:886-957, :1140-1161) --
//      the only state it touches is the small per-aircraft gate record;
//   2. the global CPR decode of the batch's pairs (:1079-1121), a pure function of four integers and a flag: four pairs per step;
//   3. in frame order again: the aircraft update from the decoded fields (:530-672, :1124-1175; the fields come from the GPU's
//      ordering pass, or from the same function on the host, decode1090.h) and the callback for every accepted frame.
#pragma once

#include <array>
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <thread>
#include <vector>

#include "adsb_amd.h"

namespace adsb_amd
{

int cpr_nl(double lat);
// Global airborne CPR from an even and an odd frame; false when the two latitudes fall in different NL zones.
bool cpr_global(int32_t even_lat, int32_t even_lon, int32_t odd_lat, int32_t odd_lon, bool use_even, int32_t* lat1e7, int32_t* lon1e7); // raw 17-bit CPR values

// The same for n pairs held as structure of arrays (AVX2, four pairs per step): ok[i] = 1 and lat1e7[i], lon1e7[i] written where pair i
// decodes, ok[i] = 0 (outputs untouched) where cpr_global would return false.  Identical results, pair by pair.
void cpr_global_batch(size_t n, const int32_t* even_lat, const int32_t* even_lon, const int32_t* odd_lat, const int32_t* odd_lon, const uint8_t* use_even,
                      int32_t* lat1e7, int32_t* lon1e7, uint8_t* ok);

class Resolver1090
{
  public:
    Resolver1090();
    ~Resolver1090();
    Resolver1090(const Resolver1090&)            = delete;
    Resolver1090& operator=(const Resolver1090&) = delete;
    // rate_hz == 0: wall clock like the reference; otherwise the stream time of the sample.
    void   set_sample_clock(int64_t t0_ns, uint32_t rate_hz);
    void   set_mode(int samples_per_us_x10) { per_us_x10_ = samples_per_us_x10 == 24 ? 24 : 20; }
    // `dec` (parallel to `rec`): the GPU's decoded fields (adsb_amd_scan_1090_fetch_decoded); NULL: decode on the host (decode1090.h).
    long   feed(const adsb_amd_record_t* rec, const adsb_amd_decoded_t* dec, size_t n, size_t samples_per_buffer, size_t nbuffers,
                adsb_amd_on_changed_fn cb, void* user);
    // the same from the packed hand-over form (no message bytes: the frames passed to the callback have msg all zero)
    long   feed_packed(const adsb_amd_packed_t* packed, size_t n, size_t samples_per_buffer, size_t nbuffers, adsb_amd_on_changed_fn cb, void* user);
    size_t aircraft_count() const { return table_.size(); }
    // Added to the `offset` of every frame handed to the callback by the next feed: a caller that cuts one delivery into several feeds (the
    // recorded-file replay's batches) keeps the frames' sample indices those of the whole delivery.
    void   set_frame_offset_base(uint64_t samples) { frame_base_ = samples; }

  private:
    // What the sequential pass reads and writes per aircraft: the reference's ICAO cache entry (:195-207) and the raw halves of the
    // CPR pair with their times (:1140-1161).  [0] = even, [1] = odd; time 0 = never (the reference's default time_point).
    struct Gate
    {
        int64_t seen_ns   = 0; // last clean DF11/17; valid when `seen`
        int64_t pos_ns[2] = {0, 0};
        int32_t lat[2]    = {0, 0};
        int32_t lon[2]    = {0, 0};
        bool    seen      = false;
    };
    // The reference keeps two unordered_maps keyed by the 24-bit address (ICAO cache :195-207, TrafficManager's aircraft,
    // AircraftImpl.h:49-68).  Every address enters both at the same moment (a clean DF11/17 is accepted in the same step
    // that whitelists it), so one open-addressing table serves both: one probe per frame instead of two hash look-ups.
    // The probe array holds only {address + 1, index} (8 bytes a slot, a few KiB for a busy sky, resident in L1); the gate
    // state and the published aircraft live in two arrays by that index (the sequential pass touches only the first).
    class AddrTable
    {
      public:
        AddrTable() : slots_(1024) {}
        int32_t find(uint32_t addr) const
        {
            const uint32_t key = addr + 1u;
            for (size_t i = hash(addr) & (slots_.size() - 1);; i = (i + 1) & (slots_.size() - 1))
            {
                if (slots_[i].key == key) return (int32_t)slots_[i].index;
                if (slots_[i].key == 0) return -1;
            }
        }
        // the address is known to be absent
        uint32_t insert(uint32_t addr)
        {
            if ((count_ + 1) * 2 > slots_.size()) grow();
            size_t i = hash(addr) & (slots_.size() - 1);
            while (slots_[i].key) i = (i + 1) & (slots_.size() - 1);
            slots_[i].key   = addr + 1u;
            slots_[i].index = (uint32_t)count_;
            return (uint32_t)count_++;
        }
        size_t size() const { return count_; }

      private:
        struct Slot
        {
            uint32_t key = 0; // address + 1; 0 = empty (addresses are 24 bits, the AP xor parity candidates at most 24 as well)
            uint32_t index = 0;
        };
        static size_t hash(uint32_t a) { return (size_t)((a * 0x9E3779B1u) >> 8); }
        void          grow()
        {
            std::vector<Slot> old;
            old.swap(slots_);
            slots_.assign(old.size() * 2, Slot{});
            for (const Slot& s : old)
                if (s.key)
                {
                    size_t i = hash(s.key - 1u) & (slots_.size() - 1);
                    while (slots_[i].key) i = (i + 1) & (slots_.size() - 1);
                    slots_[i] = s;
                }
        }
        std::vector<Slot> slots_;
        size_t            count_ = 0;
    };
    struct Block; // one batch of accepted frames between the sequential pass and the update pass
    // where the sequential pass stands in the caller's record array
    struct Walk
    {
        size_t   i           = 0;
        uint32_t cur_buffer  = 0xFFFFFFFFu;
        uint64_t next_offset = 0; // first offset of the current buffer the reference's loop would still look at
        uint64_t buf_rem     = 0;
        int64_t  buf_t       = 0; // t0 + the whole seconds of the buffer's first sample
        int64_t  wall        = 0; // wall-clock mode: the call's time
        long     accepted    = 0;
    };
    struct Job
    {
        const adsb_amd_record_t*  rec = nullptr; // records (+ dec, or decoded here), or
        const adsb_amd_decoded_t* dec = nullptr;
        const adsb_amd_packed_t*  pk  = nullptr; // the packed form
        size_t                    n = 0, samples_per_buffer = 0;
    };
    template <int SRC>
    void gate_pass(Block& blk, Walk& w, const Job& job);
    void gate_dispatch(Block& blk, Walk& w, const Job& job);
    long run(const Job& job, size_t nbuffers, adsb_amd_on_changed_fn cb, void* user);
    void update_pass(Block& blk, const Job& job, adsb_amd_on_changed_fn cb, void* user);
    void helper_main();

    AddrTable                        table_;
    std::vector<Gate>                gates_;
    std::vector<adsb_amd_aircraft_t> pubs_;
    // Two threads on a large call: a helper runs the sequential pass ahead, block by block, the caller's thread decodes the pairs,
    // updates the aircraft and fires the callbacks (in order, on the thread that called feed()).  A ring of blocks between them.
    static constexpr size_t kRing        = 4;
    static constexpr size_t kParallelMin = 8192; // records; below that a call is done on the caller's thread alone
    Block*                  blocks_      = nullptr; // kRing of them
    std::thread             helper_;
    int                     helper_cpu_ = -1; // CPU the helper pins itself to (one that shares the caller's last-level cache), -1: none
    std::mutex              m_;
    std::condition_variable cv_;
    bool                    quit_ = false, job_posted_ = false;
    Job                     job_;
    std::atomic<uint64_t>   produced_{0}, consumed_{0};
    std::atomic<bool>       gate_done_{false};
    int64_t   t0_ns_       = 0;
    uint32_t  rate_hz_     = 0;
    uint64_t  ns_per_sample_ = 0; // 10^9 / rate when that is a whole number (2 MS/s: 500), else 0
    uint64_t  rate_recip_  = 0;   // floor(2^64 / rate): quotient estimate for the exact division by the rate
    uint64_t  stream_base_ = 0;
    uint64_t  frame_base_  = 0;
    uint32_t  per_us_x10_  = 20; // samples per microsecond x 10 of the records (skip-ahead length)
};

} // namespace adsb_amd
