// resolver1090.hpp -- host half of the 1090 path: candidate records -> accepted frames -> aircraft state.
//
// The GPU emits every (offset, pass) the reference *could* accept; what it *does* accept depends on state that is
// inherently sequential (reference ADSB1090.cpp:886-957): a frame accepted at j hides the next 128/240 offsets,
// AP-type DFs are valid only if their address was recently seen in a clean DF11/17 (:195-207, :396-435), and the
// retry slice is only looked at when the first slice was not accepted.  Resolver1090 walks the sorted records once
// and applies exactly those rules, then decodes the fields the aircraft update consumes (:530-672), runs the
// global CPR decode (:1079-1121) and fires the callback for every accepted frame (:1124-1175).
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "adsb_amd.h"

namespace adsb_amd
{

// Fields of one decoded Mode S message that the aircraft update uses (the reference's `Message`, ADSB1090.cpp:32-97).
struct ModesFields
{
    int      df = 0, nbits = 0, errorbit = -1;
    uint32_t icao   = 0;
    int      metype = 0, mesub = 0;
    bool     odd      = false; // CPR format flag
    int      raw_lat  = 0, raw_lon = 0;
    int      altitude = 0;
    int      velocity = 0, heading = 0;
    int      identity = 0; // squawk as four decimal digits
    std::array<char, 8> flight{};
};

ModesFields decode_fields(const uint8_t msg[14], int df, int nbits, int errorbit, uint32_t icao);
int         cpr_nl(double lat);
// Global airborne CPR from an even and an odd frame; false when the two latitudes fall in different NL zones.
bool cpr_global(double even_lat, double even_lon, double odd_lat, double odd_lon, bool use_even, int32_t* lat1e7, int32_t* lon1e7);

class Resolver1090
{
  public:
    // rate_hz == 0: wall clock like the reference; otherwise the stream time of the sample.
    void   set_sample_clock(int64_t t0_ns, uint32_t rate_hz);
    long   feed(const adsb_amd_record_t* rec, size_t n, size_t samples_per_buffer, size_t nbuffers, adsb_amd_on_changed_fn cb, void* user);
    size_t aircraft_count() const { return table_.size(); }

  private:
    struct Track
    {
        adsb_amd_aircraft_t pub{};
        double              even_lat = 0, even_lon = 0, odd_lat = 0, odd_lon = 0;
        int64_t             even_ns = 0, odd_ns = 0; // 0 = never (the reference's default time_point)
        int64_t             seen_ns = 0;             // last clean DF11/17 (the reference's ICAO cache entry, :195-207)
        bool                seen    = false;
    };
    // The reference keeps two unordered_maps keyed by the 24-bit address (ICAO cache :195-207, TrafficManager's aircraft,
    // AircraftImpl.h:49-68).  Every address enters both at the same moment (a clean DF11/17 is accepted in the same step
    // that whitelists it), so one open-addressing table serves both: one probe per frame instead of two hash look-ups.
    class AddrTable
    {
      public:
        AddrTable() : slots_(1024) {}
        Track* find(uint32_t addr)
        {
            for (size_t i = hash(addr) & (slots_.size() - 1);; i = (i + 1) & (slots_.size() - 1))
            {
                if (!slots_[i].used) return nullptr;
                if (slots_[i].addr == addr) return &slots_[i].track;
            }
        }
        Track& get_or_create(uint32_t addr, bool* created)
        {
            if ((count_ + 1) * 2 > slots_.size()) grow();
            for (size_t i = hash(addr) & (slots_.size() - 1);; i = (i + 1) & (slots_.size() - 1))
            {
                if (!slots_[i].used)
                {
                    slots_[i].used = true;
                    slots_[i].addr = addr;
                    count_++;
                    *created = true;
                    return slots_[i].track;
                }
                if (slots_[i].addr == addr)
                {
                    *created = false;
                    return slots_[i].track;
                }
            }
        }
        size_t size() const { return count_; }

      private:
        struct Slot
        {
            uint32_t addr = 0;
            bool     used = false;
            Track    track;
        };
        static size_t hash(uint32_t a) { return (size_t)((a * 0x9E3779B1u) >> 8); }
        void          grow()
        {
            std::vector<Slot> old;
            old.swap(slots_);
            slots_.resize(old.size() * 2);
            count_ = 0;
            for (Slot& s : old)
                if (s.used)
                {
                    bool c;
                    get_or_create(s.addr, &c) = s.track;
                }
        }
        std::vector<Slot> slots_;
        size_t            count_ = 0;
    };
    int64_t now_ns(uint64_t stream_sample) const;
    void    apply(const ModesFields& f, int64_t t, Track& a);

    AddrTable                             table_;
    int64_t                               t0_ns_       = 0;
    uint32_t                              rate_hz_     = 0;
    uint64_t                              stream_base_ = 0;
};

} // namespace adsb_amd
