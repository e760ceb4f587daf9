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
#include <unordered_map>

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
    size_t aircraft_count() const { return aircraft_.size(); }

  private:
    struct Track
    {
        adsb_amd_aircraft_t pub{};
        double              even_lat = 0, even_lon = 0, odd_lat = 0, odd_lon = 0;
        int64_t             even_ns = 0, odd_ns = 0; // 0 = never (the reference's default time_point)
    };
    int64_t now_ns(uint64_t stream_sample) const;
    void    apply(const ModesFields& f, int64_t t, Track& a);

    std::unordered_map<uint32_t, int64_t> icao_seen_;
    std::unordered_map<uint32_t, Track>   aircraft_;
    int64_t                               t0_ns_       = 0;
    uint32_t                              rate_hz_     = 0;
    uint64_t                              stream_base_ = 0;
};

} // namespace adsb_amd
