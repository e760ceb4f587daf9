/*
 * synth978.c -- deterministic synthetic UAT 978 u8 IQ generator (test + bench input; neither product path nor oracle).
 *
 * UAT: 1.041667 Mbit/s continuous-phase FSK, modulation index 0.6 (+-312.5 kHz), sampled at 2.083334 MS/s = 2 samples per
 * bit (reference UAT978.cpp:29).  A frame is the 36-bit sync word (0xEACDDA4E2 downlink / 0x153225B1D uplink) followed by
 * 240 (short) or 384 (long) bits of RS(30,18) / RS(48,34) code word, or 4416 bits of six interleaved RS(92,72) blocks.
 * Own small Reed-Solomon encoder (GF(256) poly 0x187, first root 120) so that this file depends on nothing else.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC synth978.c -lm
 */
#include "synth978.h"

#include <math.h>
#include <string.h>

static inline uint64_t sm64_next(uint64_t* s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z          = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint32_t sm64_below(uint64_t* s, uint32_t n) { return (uint32_t)(((sm64_next(s) >> 32) * (uint64_t)n) >> 32); }

/* ---- GF(256) / RS encoder */
static uint8_t gf_exp[512], gf_log[256];
static int     gf_ready = 0;
static void    gf_init(void)
{
    if (gf_ready) return;
    int x = 1;
    for (int i = 0; i < 255; i++)
    {
        gf_exp[i] = (uint8_t)x;
        gf_log[x] = (uint8_t)i;
        x <<= 1;
        if (x & 0x100) x ^= 0x187;
    }
    for (int i = 255; i < 512; i++) gf_exp[i] = gf_exp[i - 255];
    gf_ready = 1;
}
static uint8_t gf_mul(uint8_t a, uint8_t b) { return (a && b) ? gf_exp[gf_log[a] + gf_log[b]] : 0; }

/* systematic encoder: parity = (data(x) * x^nroots) mod g(x), g(x) = prod_{i<nroots} (x - alpha^(120+i)) */
static void rs_parity(const uint8_t* data, int k, int nroots, uint8_t* parity)
{
    uint8_t g[33];
    gf_init();
    memset(g, 0, sizeof(g));
    g[0] = 1; /* g as coefficients, g[0] = x^0 term, built up to degree nroots */
    for (int i = 0; i < nroots; i++)
    {
        uint8_t root = gf_exp[(120 + i) % 255];
        for (int j = i + 1; j > 0; j--) g[j] = (uint8_t)(g[j - 1] ^ gf_mul(g[j], root));
        g[0] = gf_mul(g[0], root);
    }
    memset(parity, 0, (size_t)nroots);
    for (int i = 0; i < k; i++)
    {
        uint8_t fb = (uint8_t)(data[i] ^ parity[0]);
        for (int j = 0; j < nroots - 1; j++) parity[j] = (uint8_t)(parity[j + 1] ^ gf_mul(fb, g[nroots - 1 - j]));
        parity[nroots - 1] = gf_mul(fb, g[0]);
    }
}

void adsb_synth978_default(adsb_synth978_cfg_t* c)
{
    c->seed          = 0x978AD5BULL;
    c->noise_amp     = 3;
    c->amp_lo        = 25;
    c->amp_hi        = 110;
    c->mean_gap_bits = 3000;
    c->pct_uplink    = 5;
    c->pct_long      = 70;
    c->pct_corrupt   = 30;
    c->max_bad_bytes = 3;
}

static inline uint8_t clamp_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

long adsb_synth978_fill(const adsb_synth978_cfg_t* c, uint64_t stream_index, uint8_t* out, size_t nbytes, adsb_synth978_frame_t* frames, long cap)
{
    const size_t n = nbytes / 2;
    uint64_t     ns = c->seed * 0x9E3779B97F4A7C15ULL + stream_index * 0xD1B54A32D192ED03ULL + 1;
    /* background: carrier off, I,Q = 127/128 + small noise */
    const uint32_t span = (uint32_t)(2 * c->noise_amp + 2);
    for (size_t i = 0; i < nbytes; i++) out[i] = clamp_u8(127 - c->noise_amp + (int)sm64_below(&ns, span));

    uint64_t fs     = ns ^ 0xF4A3E5D1C0B7ULL;
    long     nfr    = 0;
    size_t   pos    = 64 + sm64_below(&fs, (uint32_t)(2 * c->mean_gap_bits)) * 2;
    static uint8_t bits[36 + 4416];
    while (pos < n)
    {
        uint8_t  cw[552];
        uint8_t  data[432];
        int      kind, nbits;
        uint32_t sel = sm64_below(&fs, 100);
        if (sel < (uint32_t)c->pct_uplink)
        {
            kind = 2;
            for (int i = 0; i < 432; i++) data[i] = (uint8_t)sm64_next(&fs);
            for (int b = 0; b < 6; b++)
            {
                uint8_t blk[92];
                memcpy(blk, data + 72 * b, 72);
                rs_parity(blk, 72, 20, blk + 72);
                for (int i = 0; i < 92; i++) cw[i * 6 + b] = blk[i];
            }
            nbits = 4416;
        }
        else if (sm64_below(&fs, 100) < (uint32_t)c->pct_long)
        {
            kind = 1;
            for (int i = 0; i < 34; i++) data[i] = (uint8_t)sm64_next(&fs);
            data[0] = (uint8_t)(((1 + sm64_below(&fs, 10)) << 3) | (data[0] & 7)); /* MDB type 1..10 */
            memcpy(cw, data, 34);
            rs_parity(cw, 34, 14, cw + 34);
            nbits = 384;
        }
        else
        {
            kind = 0;
            for (int i = 0; i < 18; i++) data[i] = (uint8_t)sm64_next(&fs);
            data[0] &= 7; /* MDB type 0 */
            memcpy(cw, data, 18);
            rs_parity(cw, 18, 12, cw + 18);
            nbits = 240;
        }
        int bad = 0;
        if (sm64_below(&fs, 100) < (uint32_t)c->pct_corrupt && c->max_bad_bytes > 0)
        {
            bad = 1 + (int)sm64_below(&fs, (uint32_t)c->max_bad_bytes);
            for (int e = 0; e < bad; e++)
            {
                uint32_t at = sm64_below(&fs, (uint32_t)(nbits / 8));
                cw[at] ^= (uint8_t)(1 + sm64_below(&fs, 255));
            }
        }
        const uint64_t sync = (kind == 2) ? 0x153225B1DULL : 0xEACDDA4E2ULL;
        for (int i = 0; i < 36; i++) bits[i] = (uint8_t)((sync >> (35 - i)) & 1);
        for (int i = 0; i < nbits; i++) bits[36 + i] = (uint8_t)((cw[i >> 3] >> (7 - (i & 7))) & 1);

        const double amp  = (double)c->amp_lo + (double)sm64_below(&fs, (uint32_t)(c->amp_hi - c->amp_lo + 1));
        double       phi  = 2.0 * M_PI * (double)sm64_below(&fs, 4096) / 4096.0;
        const double step = M_PI * 0.6 / 2.0; /* modulation index 0.6, two samples per bit */
        const int    half = (int)sm64_below(&fs, 2); /* sample grid offset of half a sample interval within the bit */
        (void)half;
        const int total = (36 + nbits) * 2;
        for (int s = 0; s <= total && pos + (size_t)s < n; s++)
        {
            size_t o   = 2 * (pos + (size_t)s);
            out[o]     = clamp_u8((int)lround(127.5 + amp * cos(phi)) + (int)out[o] - 127);
            out[o + 1] = clamp_u8((int)lround(127.5 + amp * sin(phi)) + (int)out[o + 1] - 127);
            if (s < total) phi += bits[s >> 1] ? step : -step;
        }
        if (frames && nfr < cap)
        {
            adsb_synth978_frame_t* f = &frames[nfr];
            f->start                 = pos;
            f->kind                  = (uint8_t)kind;
            f->bad_bytes             = (uint8_t)bad;
            f->len                   = (uint16_t)(kind == 2 ? 432 : kind == 1 ? 34 : 18);
            memcpy(f->data, data, f->len);
        }
        nfr++;
        pos += (size_t)total + 64 + (size_t)sm64_below(&fs, (uint32_t)(2 * c->mean_gap_bits)) * 2;
    }
    return nfr;
}
