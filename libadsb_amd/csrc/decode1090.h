// decode1090.h -- the stateless part of DecodeModesMessage (reference ADSB1090.cpp:491-675): the fields the aircraft update
// (InteractiveReceiveData, :1124-1175) consumes, as a pure function of the 14 message bytes and the DF.
//
// One source, two builds: the ordering pass of the GPU scan runs it per record (scan1090.hip, gather_sorted_kernel) so that
// the host's sequential pass only gates, sequences and updates aircraft state; the resolver's plain record path and the CPU
// tests run the host build.  Both must give the same bytes:
//   * integer fields are bit fiddling;
//   * velocity = (int)sqrt((double)(ns^2 + ew^2)) (:645) is the integer square root (n < 2^21: a correctly rounded double
//     square root truncates to it), computed here with an integer fix-up so that no libm is trusted;
//   * heading = (int)(atan2(ew, ns) * 360 / (2 pi)), truncated toward zero, +360 when negative (:654-659).  For |ew|, |ns| <= 1023
//     the angle in degrees is an exact integer only on the eight directions ew = 0, ns = 0, |ew| = |ns| (tan of a whole number of
//     degrees is rational only at multiples of 45); there the host's libm gives exactly 0, +-45, +-90, +-135, 180 and both builds
//     return those constants without calling atan2.  Everywhere else the distance to the nearest integer is at least 1.3e-6 degrees
//     (exhaustive over the 2047 x 2047 lattice, tests/test_capi_cpu.py), nine orders of magnitude above the error of any
//     double-precision atan2, so the truncation cannot differ between the host's and the device's math library.
#pragma once

#include <math.h>
#include <stdint.h>

#include "adsb_amd.h"

#if defined(__HIPCC__)
#define ADSB_AMD_HD __host__ __device__
#else
#define ADSB_AMD_HD
#endif

namespace adsb_amd
{

// 13-bit AC field with M=0,Q=1 -> feet (:440-466); anything else reports 0
ADSB_AMD_HD inline int altitude_ac13(const uint8_t* g)
{
    const bool metric = (g[3] & 0x40) != 0, q = (g[3] & 0x10) != 0;
    if (metric || !q) return 0;
    const int n = ((g[2] & 0x1F) << 6) | ((g[3] & 0x80) >> 2) | ((g[3] & 0x20) >> 1) | (g[3] & 0x0F);
    return n * 25 - 1000;
}
// 12-bit AC field of the airborne position message (:470-486)
ADSB_AMD_HD inline int altitude_ac12(const uint8_t* g)
{
    if ((g[5] & 1) == 0) return 0;
    const int n = ((g[5] >> 1) << 4) | (g[6] >> 4);
    return n * 25 - 1000;
}
// 6-bit AIS character set of the identification message (:608)
ADSB_AMD_HD inline char ais_char(unsigned v)
{
    v &= 63u;
    if (v >= 1 && v <= 26) return (char)('A' + (v - 1));
    if (v == 32) return ' ';
    if (v >= 48 && v <= 57) return (char)('0' + (v - 48));
    return '?';
}
ADSB_AMD_HD inline int isqrt21(int n)
{ // floor(sqrt(n)), 0 <= n < 2^21
#if defined(__HIP_DEVICE_COMPILE__)
    int r = (int)__builtin_sqrtf((float)n); // any start within a few units will do: the two loops below make it exact
#else
    int r = (int)sqrt((double)n);
#endif
    while (r * r > n) r--;
    while ((r + 1) * (r + 1) <= n) r++;
    return r;
}
// (int)(atan2(ew, ns) * 360 / (2 pi)) with the reference's wrap (:654-659); see the header comment
ADSB_AMD_HD inline int heading_of(int ewv, int nsv)
{
    int h;
    const int aew = ewv < 0 ? -ewv : ewv, ans = nsv < 0 ? -nsv : nsv;
    if (ewv == 0) h = nsv >= 0 ? 0 : 180;
    else if (nsv == 0) h = ewv > 0 ? 90 : -90;
    else if (aew == ans) h = ewv > 0 ? (nsv > 0 ? 45 : 135) : (nsv > 0 ? -45 : -135);
    else
    {
#if defined(__HIP_DEVICE_COMPILE__)
        // Device build: the single-precision angle is within 1e-4 degrees of the true one (a few ulp of atan2f at <= 180, one more
        // rounding in the multiply), the true one is never closer than 1.3e-6 degrees to a whole number (header), so wherever the
        // estimate is more than 1e-3 away from a whole number it truncates to the same integer.  Only the rest (two velocity
        // frames in a thousand) pays for the double-precision atan2, which cost the ordering pass 5 of its 14 us when every wave ran it.
        const float a = atan2f((float)ewv, (float)nsv) * 57.295779513f;
        const float f = __builtin_fabsf(a) - __builtin_floorf(__builtin_fabsf(a));
        if (f > 1e-3f && f < 1.0f - 1e-3f) h = (int)a;
        else
#endif
        h = (int)(atan2((double)ewv, (double)nsv) * 360 / (3.14159265358979323846 * 2));
    }
    return h < 0 ? h + 360 : h;
}

// kinds of aircraft update a message leads to (InteractiveReceiveData :1124-1175)
enum : uint8_t
{
    ADSB_AMD_K_NONE     = 0, // DF11, DF5/21/16/24, DF17 types that update nothing
    ADSB_AMD_K_ALTITUDE = 1, // DF0/4/20: altitude from the AC13 field
    ADSB_AMD_K_IDENT    = 2, // DF17 type 1-4: callsign
    ADSB_AMD_K_POSITION = 3, // DF17 type 9-18: altitude (AC12) + raw CPR latitude/longitude + format flag
    ADSB_AMD_K_VELOCITY = 4, // DF17 type 19 subtype 1-2: speed + track
};

ADSB_AMD_HD inline adsb_amd_decoded_t decode_record(const uint8_t* g, int df)
{
    adsb_amd_decoded_t d;
    d.kind     = ADSB_AMD_K_NONE;
    d.metype   = (uint8_t)(g[4] >> 3);
    d.mesub    = (uint8_t)(g[4] & 7);
    d.odd      = 0;
    d.altitude = 0;
    d.a        = 0;
    d.b        = 0;
    if (df == 0 || df == 4 || df == 20)
    {
        d.kind     = ADSB_AMD_K_ALTITUDE;
        d.altitude = altitude_ac13(g); // :598
        return d;
    }
    if (df != 17) return d;
    const int metype = d.metype, mesub = d.mesub;
    if (metype >= 1 && metype <= 4)
    { // eight 6-bit characters in bytes 5..10 (:612-619), packed first character = lowest byte
        uint64_t v = 0;
        for (int i = 5; i <= 10; i++) v = (v << 8) | g[i];
        uint32_t lo = 0, hi = 0;
        for (int i = 0; i < 4; i++) lo |= (uint32_t)(uint8_t)ais_char((unsigned)(v >> (42 - 6 * i))) << (8 * i);
        for (int i = 4; i < 8; i++) hi |= (uint32_t)(uint8_t)ais_char((unsigned)(v >> (42 - 6 * i))) << (8 * (i - 4));
        d.kind = ADSB_AMD_K_IDENT;
        d.a    = lo;
        d.b    = hi;
    }
    else if (metype >= 9 && metype <= 18)
    { // airborne position (:622-630)
        d.kind     = ADSB_AMD_K_POSITION;
        d.odd      = (g[6] & 0x04) ? 1 : 0;
        d.altitude = altitude_ac12(g);
        d.a        = (uint32_t)(((g[6] & 3) << 15) | (g[7] << 7) | (g[8] >> 1));
        d.b        = (uint32_t)(((g[8] & 1) << 16) | (g[9] << 8) | g[10]);
    }
    else if (metype == 19 && (mesub == 1 || mesub == 2))
    { // airborne velocity over ground (:631-660)
        const int ew = ((g[5] & 3) << 8) | g[6];
        const int ns = ((g[7] & 0x7F) << 3) | (g[8] >> 5);
        const int v  = isqrt21(ns * ns + ew * ew);
        int       h  = 0;
        if (v != 0) h = heading_of((g[5] & 4) ? -ew : ew, (g[7] & 0x80) ? -ns : ns);
        d.kind = ADSB_AMD_K_VELOCITY;
        d.a    = (uint32_t)v;
        d.b    = (uint32_t)h;
    }
    return d;
}

} // namespace adsb_amd
