// rs978.h -- Reed-Solomon decoder for the three UAT codes, one source for the device (uat978.hip, one lane per code word)
// and for the host (the exported adsb_amd_uat_rs_decode, which the CPU tests hold against oracle/oracle978.c).
//
// GF(256), field polynomial 0x187, first consecutive root 120, primitive element 1; RS(30,18), RS(48,34) and RS(92,72)
// as shortened RS(255, 255 - nroots).  Berlekamp-Massey / Chien / Forney with libfec's conventions, which matter for
// words that are NOT within the correction radius: a correction located in the zero padding is dropped but still
// counted, and a zero Forney denominator is not rejected.  Everything is in polynomial (not index) form.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RS978_HD __host__ __device__ __forceinline__
#else
#define RS978_HD inline
#endif

namespace adsb_amd
{
struct RsTables
{
    uint8_t exp[512]; // alpha^i for i in [0, 510): no modulo needed after adding two logarithms
    uint8_t log[256]; // log[0] is never used
};

inline void rs978_build_tables(RsTables& t)
{
    int x = 1;
    for (int i = 0; i < 255; i++)
    {
        t.exp[i] = t.exp[i + 255] = (uint8_t)x;
        t.log[x]                  = (uint8_t)i;
        x <<= 1;
        if (x & 0x100) x ^= 0x187;
    }
    t.exp[510] = t.exp[0], t.exp[511] = t.exp[1];
    t.log[0] = 0;
}

constexpr int kRsFcr      = 120;
constexpr int kRsMaxRoots = 20;

struct RsWork // per code word scratch; on the device it lives in LDS
{
    uint8_t s[kRsMaxRoots], lambda[kRsMaxRoots + 1], b[kRsMaxRoots + 1], t[kRsMaxRoots + 1], omega[kRsMaxRoots], root[kRsMaxRoots], loc[kRsMaxRoots];
};

RS978_HD uint8_t rs_mul(const RsTables& T, uint8_t a, uint8_t b) { return (a && b) ? T.exp[T.log[a] + T.log[b]] : (uint8_t)0; }

// syndrome `i` (root alpha^(fcr + i)) of the n symbols data[0], data[stride], ...: Horner
RS978_HD uint8_t rs978_syndrome(const RsTables& T, int i, int n, const uint8_t* data, int stride)
{
    uint8_t acc = 0;
    for (int j = 0; j < n; j++) acc = (uint8_t)((acc ? T.exp[T.log[acc] + (kRsFcr + i)] : 0) ^ data[j * stride]);
    return acc;
}

RS978_HD int rs978_decode_with_syndromes(const RsTables& T, int nroots, int pad, uint8_t* data, int stride, RsWork& w);

// In place over data[0], data[stride], ... ((255 - pad) symbols).  Returns the number of located errors, or -1 with the
// data untouched.
RS978_HD int rs978_decode(const RsTables& T, int nroots, int pad, uint8_t* data, int stride, RsWork& w)
{
    for (int i = 0; i < nroots; i++) w.s[i] = rs978_syndrome(T, i, 255 - pad, data, stride);
    return rs978_decode_with_syndromes(T, nroots, pad, data, stride, w);
}

// the same with w.s[0 .. nroots) already holding the syndromes (the device computes them one per lane)
RS978_HD int rs978_decode_with_syndromes(const RsTables& T, int nroots, int pad, uint8_t* data, int stride, RsWork& w)
{
    const int nr  = nroots;
    uint8_t   any = 0;
    for (int i = 0; i < nr; i++) any |= w.s[i];
    if (!any) return 0;

    for (int i = 0; i <= nr; i++) w.lambda[i] = w.b[i] = 0;
    w.lambda[0] = w.b[0] = 1;
    int el = 0;
    for (int r = 1; r <= nr; r++)
    {
        uint8_t discr = 0;
        for (int i = 0; i < r; i++) discr ^= rs_mul(T, w.lambda[i], w.s[r - i - 1]);
        if (discr == 0)
        {
            for (int i = nr; i > 0; i--) w.b[i] = w.b[i - 1];
            w.b[0] = 0;
            continue;
        }
        w.t[0] = w.lambda[0];
        for (int i = 0; i < nr; i++) w.t[i + 1] = (uint8_t)(w.lambda[i + 1] ^ rs_mul(T, discr, w.b[i]));
        if (2 * el <= r - 1)
        {
            el = r - el;
            const int ld = 255 - T.log[discr];
            for (int i = 0; i <= nr; i++) w.b[i] = w.lambda[i] ? T.exp[T.log[w.lambda[i]] + ld] : (uint8_t)0;
        }
        else
        {
            for (int i = nr; i > 0; i--) w.b[i] = w.b[i - 1];
            w.b[0] = 0;
        }
        for (int i = 0; i <= nr; i++) w.lambda[i] = w.t[i];
    }
    int deg = 0;
    for (int i = 0; i <= nr; i++)
        if (w.lambda[i]) deg = i;

    // roots of lambda: X^-1 = alpha^i  <=>  error (i - 1) symbols before the end of the full 255-symbol word
    int count = 0;
    for (int i = 1; i <= 255 && count < deg; i++)
    {
        uint8_t q = 1;
        for (int j = 1; j <= deg; j++)
            if (w.lambda[j]) q ^= T.exp[(T.log[w.lambda[j]] + j * i) % 255];
        if (q) continue;
        w.root[count] = (uint8_t)i;
        w.loc[count]  = (uint8_t)(i - 1);
        count++;
    }
    if (count != deg) return -1;

    for (int i = 0; i < deg; i++)
    {
        uint8_t acc = 0;
        for (int j = 0; j <= i; j++) acc ^= rs_mul(T, w.s[i - j], w.lambda[j]);
        w.omega[i] = acc;
    }
    for (int j = count - 1; j >= 0; j--)
    {
        const int rt   = w.root[j];
        uint8_t   num1 = 0;
        for (int i = deg - 1; i >= 0; i--)
            if (w.omega[i]) num1 ^= T.exp[(T.log[w.omega[i]] + i * rt) % 255];
        const uint8_t num2 = T.exp[(rt * (kRsFcr - 1) + 255) % 255];
        uint8_t       den  = 0;
        const int     top  = (deg < nr - 1 ? deg : nr - 1) & ~1;
        for (int i = top; i >= 0; i -= 2)
            if (w.lambda[i + 1]) den ^= T.exp[(T.log[w.lambda[i + 1]] + i * rt) % 255];
        if (num1 != 0 && w.loc[j] >= pad)
        {
            const int lden = den ? T.log[den] : 255; // libfec: the index form of zero is 255, the exponent then gains 255 - 255
            data[(w.loc[j] - pad) * stride] ^= T.exp[(T.log[num1] + T.log[num2] + 255 - lden) % 255];
        }
    }
    return count;
}

constexpr int kUatShortSkip = 36 + 240, kUatLongSkip = 36 + 384, kUatUplinkSkip = 36 + 4416;

// correct_adsb_frame: long first, in place; then short on whatever the long attempt left.  Returns the bits to jump
// (0 = neither) and the corrected-symbol count (9999 = neither).  `short_syndromes` (12, of the first 30 bytes as they
// were before the long attempt) may be given; they are recomputed if the long attempt changed the frame.
RS978_HD int rs978_correct_adsb_with_syndromes(const RsTables& T, uint8_t* frame48, RsWork& w, const uint8_t* short_syndromes, int* rs)
{
    int n = rs978_decode_with_syndromes(T, 14, 207, frame48, 1, w); // w.s = the 14 long syndromes
    if (n >= 0 && n <= 7 && (frame48[0] >> 3) != 0)
    {
        *rs = n;
        return kUatLongSkip;
    }
    if (n > 0 || !short_syndromes)
        for (int i = 0; i < 12; i++) w.s[i] = rs978_syndrome(T, i, 30, frame48, 1);
    else
        for (int i = 0; i < 12; i++) w.s[i] = short_syndromes[i];
    n = rs978_decode_with_syndromes(T, 12, 225, frame48, 1, w);
    if (n >= 0 && n <= 6 && (frame48[0] >> 3) == 0)
    {
        *rs = n;
        return kUatShortSkip;
    }
    *rs = 9999;
    return 0;
}

RS978_HD int rs978_correct_adsb(const RsTables& T, uint8_t* frame48, RsWork& w, int* rs)
{
    for (int i = 0; i < 14; i++) w.s[i] = rs978_syndrome(T, i, 48, frame48, 1);
    return rs978_correct_adsb_with_syndromes(T, frame48, w, nullptr, rs);
}
} // namespace adsb_amd
