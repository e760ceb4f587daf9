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

// one per 18-bit match, in candidate order; variant v = frame taken from sample index + v
struct uat_adsb_rec_t
{
    uint32_t index;       // sample index of the first sync bit
    uint32_t uplink_slot; // kind 1: where the uplink record went
    uint8_t  kind;        // 0 = ADS-B sync word, 1 = uplink sync word
    uint8_t  ok[2];       // 36-bit sync re-check passed
    uint8_t  pad;
    int16_t  center[2];
    uint8_t  frame[2][kUatLongBytes];
    uint64_t window;   // sign bits of samples [2 * (index >> 1), +64): both 18-bit registers at detection time
    uint64_t after[2]; // sign bits of the 64 samples from bit (index >> 1) + 276 + 1 (short) / + 420 + 1 (long; uplink: + 4452 + 1 in [0])
};
struct uat_uplink_rec_t
{
    uint8_t ok[2];
    uint8_t pad[2];
    int16_t center[2];
    uint8_t frame[2][kUatUplinkBytes];
};

struct UatArgs
{
    const uint16_t*   in;  // u8 IQ pairs as u16 (phases_given == 0) or u16 phases (phases_given == 1)
    const uint16_t*   lut; // 65536-entry phase LUT
    uint64_t          nsamples;
    int               phases_given;
    uint64_t*         signs; // ceil(nsamples / 64) + 2 words
    uint32_t*         cand;
    uint32_t          cand_cap;
    uint32_t*         counts; // [0] candidates, [1] uplink records
    uat_adsb_rec_t*   adsb;   // cand_cap entries
    uat_uplink_rec_t* uplink;
    uint32_t          uplink_cap;
};

hipError_t launch_uat978(const UatArgs& a, hipStream_t stream);                       // signs + 18-bit match
hipError_t launch_uat978_demod(const UatArgs& a, uint32_t ncand, hipStream_t stream); // one wave per candidate
} // namespace adsb_amd
