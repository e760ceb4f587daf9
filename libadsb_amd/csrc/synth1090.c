/*
 * synth1090.c -- deterministic synthetic 1090ES u8 IQ generator (bench + test input).
 *
 * Implements the input specification of SURVEY.md section 8(d): splitmix64 streams keyed by
 * (seed + buffer index), so any shard of a large recording can be regenerated independently
 * on any rank / box.  Signal model follows the pulse layout the reference demodulator expects
 * (reference ADSB1090.cpp:749-771): 2 samples per microsecond, preamble pulses at samples
 * 0,2,7,9, data bit b in samples 16+2b / 17+2b with '1' = high-then-low.
 *
 * This file is neither the product path nor the oracle: it only manufactures input bytes and,
 * optionally, a manifest of what was injected (used by tests as a sanity cross-check).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC -pthread synth1090.c -lm
 */
#include "synth1090.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- PRNG */
static inline uint64_t sm64_next(uint64_t* s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z          = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t sm64_hash(uint64_t x)
{
    uint64_t s = x;
    return sm64_next(&s);
}
/* uniform integer in [0, n) from the top 32 bits (multiply-shift; n < 2^31) */
static inline uint32_t sm64_below(uint64_t* s, uint32_t n)
{
    return (uint32_t)(((sm64_next(s) >> 32) * (uint64_t)n) >> 32);
}

/* ---------------------------------------------------------------- Mode S parity (CRC-24, generator 0x1FFF409) */
static uint32_t modes_crc24(const uint8_t* msg, int nbits)
{
    /* remainder of the first nbits-24 data bits times x^24 */
    uint32_t rem = 0;
    for (int i = 0; i < nbits - 24; i++)
    {
        uint32_t bit = (msg[i >> 3] >> (7 - (i & 7))) & 1u;
        uint32_t top = ((rem >> 23) & 1u) ^ bit;
        rem          = (rem << 1) & 0xFFFFFFu;
        if (top) rem ^= 0xFFF409u;
    }
    return rem;
}

void adsb_synth_default(adsb_synth_cfg_t* c)
{
    c->seed           = 0x1090AD5BULL;
    c->noise_amp      = 3;
    c->mean_spacing   = 2000;
    c->amp_lo         = 20;
    c->amp_hi         = 120;
    c->pct_df17       = 60;
    c->pct_df11       = 25;
    c->pct_bitflip    = 10;
    c->pct_halfsample = 10;
    c->pool_size      = 256;
}

uint32_t adsb_synth_pool_addr(const adsb_synth_cfg_t* c, uint32_t k)
{
    uint32_t a = (uint32_t)(sm64_hash(c->seed * 0x100000001B3ULL + 0xA1C0u + k) & 0xFFFFFFu);
    return a ? a : 0x00ABCDu;
}

/* ---------------------------------------------------------------- CPR encoding (airborne, 17 bit) */
static int cpr_nl(double lat)
{
    if (lat < 0) lat = -lat;
    if (lat >= 87.0) return lat > 87.0 ? 1 : 2;
    if (lat == 0) return 59;
    double a = 1.0 - cos(M_PI / 30.0);
    double b = cos(M_PI / 180.0 * lat);
    double v = 1.0 - a / (b * b);
    if (v < -1.0) v = -1.0;
    return (int)floor(2.0 * M_PI / acos(v));
}
static double pmod(double a, double b)
{
    double r = fmod(a, b);
    return r < 0 ? r + b : r;
}
static void cpr_encode(double lat, double lon, int odd, uint32_t* yz, uint32_t* xz)
{
    double dlat = 360.0 / (60 - odd);
    double y    = floor(131072.0 * pmod(lat, dlat) / dlat + 0.5);
    double rlat = dlat * (y / 131072.0 + floor(lat / dlat));
    int    nl   = cpr_nl(rlat) - odd;
    double dlon = 360.0 / (nl > 0 ? nl : 1);
    double x    = floor(131072.0 * pmod(lon, dlon) / dlon + 0.5);
    *yz         = ((uint32_t)y) & 0x1FFFFu;
    *xz         = ((uint32_t)x) & 0x1FFFFu;
}

static int ais_code(char ch)
{
    if (ch >= 'A' && ch <= 'Z') return ch - 'A' + 1;
    if (ch >= '0' && ch <= '9') return ch - '0' + 48;
    return 32;
}

/* Build one frame; returns nbits (56 or 112). */
static int build_frame(const adsb_synth_cfg_t* c, uint64_t* rs, uint64_t buf_index, uint8_t msg[14])
{
    memset(msg, 0, 14);
    uint32_t k    = sm64_below(rs, (uint32_t)(c->pool_size > 0 ? c->pool_size : 1));
    uint32_t icao = adsb_synth_pool_addr(c, k);
    uint32_t sel  = sm64_below(rs, 100);
    uint64_t r    = sm64_next(rs);
    int      nbits;
    int      ap_type = 0;

    /* per-aircraft slowly varying state (pure function of k and buffer index) */
    double   t    = (double)buf_index * 0.065536; /* seconds of stream at 2 MS/s */
    double   lat  = -60.0 + 120.0 * ((double)((k * 2654435761u) >> 8) / 16777216.0) + 1e-4 * t;
    double   lon  = -180.0 + 360.0 * ((double)((k * 40503u + 977u) & 0xFFFFu) / 65536.0) + 2e-4 * t;
    int      altn = (int)(40 + (k * 37u) % 1500u); /* (alt+1000)/25, 11 bit */
    uint32_t vew  = 40 + (k * 13u) % 500u;
    uint32_t vns  = 25 + (k * 29u) % 500u;

    if (sel < (uint32_t)c->pct_df17)
    {
        nbits  = 112;
        msg[0] = (17u << 3) | 5u;
        msg[1] = (uint8_t)(icao >> 16);
        msg[2] = (uint8_t)(icao >> 8);
        msg[3] = (uint8_t)icao;
        uint32_t kind = (uint32_t)(r % 10u);
        if (kind < 2)
        { /* identification, type code 4 */
            char cs[9];
            cs[0] = 'S'; cs[1] = 'Y'; cs[2] = 'N';
            cs[3] = (char)('0' + (k / 100) % 10); cs[4] = (char)('0' + (k / 10) % 10); cs[5] = (char)('0' + k % 10);
            cs[6] = (char)('A' + k % 26); cs[7] = ' '; cs[8] = 0;
            uint64_t bits48 = 0;
            for (int i = 0; i < 8; i++) bits48 = (bits48 << 6) | (uint64_t)ais_code(cs[i]);
            msg[4] = (4u << 3) | 0u;
            for (int i = 0; i < 6; i++) msg[5 + i] = (uint8_t)(bits48 >> (40 - 8 * i));
        }
        else if (kind < 8)
        { /* airborne position, type code 11, even/odd */
            int      odd = (int)((r >> 8) & 1u);
            uint32_t yz, xz;
            cpr_encode(lat, lon, odd, &yz, &xz);
            uint32_t alt12 = (((uint32_t)altn & 0x7F0u) << 1) | 0x10u | ((uint32_t)altn & 0xFu);
            msg[4]  = (11u << 3);
            msg[5]  = (uint8_t)(alt12 >> 4);
            msg[6]  = (uint8_t)(((alt12 & 0xFu) << 4) | ((uint32_t)odd << 2) | ((yz >> 15) & 3u));
            msg[7]  = (uint8_t)(yz >> 7);
            msg[8]  = (uint8_t)(((yz & 0x7Fu) << 1) | ((xz >> 16) & 1u));
            msg[9]  = (uint8_t)(xz >> 8);
            msg[10] = (uint8_t)xz;
        }
        else
        { /* airborne velocity, type code 19 subtype 1 */
            uint32_t dew = (uint32_t)((r >> 9) & 1u), dns = (uint32_t)((r >> 10) & 1u);
            msg[4]  = (19u << 3) | 1u;
            msg[5]  = (uint8_t)((dew << 2) | ((vew >> 8) & 3u));
            msg[6]  = (uint8_t)vew;
            msg[7]  = (uint8_t)((dns << 7) | ((vns >> 3) & 0x7Fu));
            msg[8]  = (uint8_t)(((vns & 7u) << 5) | 0x08u | 0x01u);
            msg[9]  = (uint8_t)(0x40u);
            msg[10] = 0x17;
        }
    }
    else if (sel < (uint32_t)(c->pct_df17 + c->pct_df11))
    {
        nbits  = 56;
        msg[0] = (11u << 3) | 5u;
        msg[1] = (uint8_t)(icao >> 16);
        msg[2] = (uint8_t)(icao >> 8);
        msg[3] = (uint8_t)icao;
    }
    else
    {
        static const uint8_t dfs[4] = {4, 5, 20, 21};
        uint32_t             df     = dfs[r & 3u];
        ap_type                     = 1;
        nbits                       = (df >= 16) ? 112 : 56;
        msg[0]                      = (uint8_t)((df << 3) | ((r >> 4) & 1u)); /* FS 0/1 */
        msg[1]                      = 0;
        if (df == 4 || df == 20)
        { /* AC13, feet, Q=1 */
            uint32_t n = (uint32_t)altn;
            msg[2]     = (uint8_t)((n >> 6) & 0x1Fu);
            msg[3]     = (uint8_t)((((n >> 5) & 1u) << 7) | (((n >> 4) & 1u) << 5) | 0x10u | (n & 15u));
        }
        else
        { /* identity: 13 arbitrary bits with the zero bit clear */
            uint32_t id = (uint32_t)((r >> 16) & 0x1FFFu) & ~0x40u;
            msg[2]      = (uint8_t)((id >> 8) & 0x1Fu);
            msg[3]      = (uint8_t)id;
        }
        if (nbits == 112)
        {
            uint64_t mb = sm64_next(rs);
            for (int i = 0; i < 7; i++) msg[4 + i] = (uint8_t)(mb >> (8 * i));
        }
    }
    uint32_t crc = modes_crc24(msg, nbits);
    if (ap_type) crc ^= icao;
    int last      = nbits / 8 - 1;
    msg[last - 2] = (uint8_t)(crc >> 16);
    msg[last - 1] = (uint8_t)(crc >> 8);
    msg[last]     = (uint8_t)crc;
    return nbits;
}

static inline uint8_t clamp_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* rate_x10: samples per microsecond times ten.  20 = the reference's 2 samples per microsecond (one pulse = one sample, ADSB1090.cpp:749-771).
 * 24 = the same pulse train (0.5 us pulses) integrated over sample bins of 1/2.4 us, the frame starting at a random fraction of a
 * sample: what a 2.4 MS/s receiver delivers.  Nothing in the reference demodulates that rate; it is a throughput workload. */
static int fill_impl(const adsb_synth_cfg_t* c, uint64_t buf_index, uint8_t* out, size_t nbytes, adsb_synth_frame_t* frames, int cap, int rate_x10)
{
    const size_t n = nbytes / 2;
    /* ---- background: I,Q = 127 + u, u uniform integer in [-A, +A] (16-bit multiply-shift per draw) */
    uint64_t       ns   = sm64_hash(c->seed + buf_index);
    const uint32_t span = (uint32_t)(2 * c->noise_amp + 1);
    size_t         i    = 0;
    while (i + 4 <= nbytes)
    {
        uint64_t r = sm64_next(&ns);
        for (int q = 0; q < 4; q++)
        {
            int u    = (int)((((uint32_t)(r >> (16 * q)) & 0xFFFFu) * span) >> 16) - c->noise_amp;
            out[i++] = clamp_u8(127 + u);
        }
    }
    if (i < nbytes)
    {
        uint64_t r = sm64_next(&ns);
        for (int q = 0; i < nbytes; q++)
        {
            int u    = (int)((((uint32_t)(r >> (16 * q)) & 0xFFFFu) * span) >> 16) - c->noise_amp;
            out[i++] = clamp_u8(127 + u);
        }
    }
    if (c->mean_spacing <= 0) return 0;

    /* ---- frames */
    uint64_t fs      = sm64_hash((c->seed + buf_index) ^ 0xF4A3E5D1C0B7ULL);
    int      nframes = 0;
    int      gapmax  = 2 * (c->mean_spacing - 240);
    if (gapmax < 32) gapmax = 32;
    size_t pos = 16 + sm64_below(&fs, (uint32_t)gapmax);
    while (pos < n)
    {
        uint8_t msg[14];
        int     nbits = build_frame(c, &fs, buf_index, msg);
        int     flip  = -1;
        if (sm64_below(&fs, 100) < (uint32_t)c->pct_bitflip)
        {
            flip = (int)sm64_below(&fs, (uint32_t)nbits);
            msg[flip >> 3] ^= (uint8_t)(0x80u >> (flip & 7));
        }
        int    half  = sm64_below(&fs, 100) < (uint32_t)c->pct_halfsample;
        double amp   = (double)c->amp_lo + (double)sm64_below(&fs, (uint32_t)(c->amp_hi - c->amp_lo + 1));
        double phi   = 2.0 * M_PI * (double)sm64_below(&fs, 4096) / 4096.0;
        double cphi  = cos(phi), sphi = sin(phi);
        int    nsamp = (8 + nbits) * 2;

        float level[300];
        memset(level, 0, sizeof(level));
        static const int pre[4] = {0, 2, 7, 9};
        if (rate_x10 == 20)
        {
            for (int p = 0; p < 4; p++)
            {
                if (half) { level[pre[p]] += 0.5f; level[pre[p] + 1] += 0.5f; }
                else level[pre[p]] += 1.0f;
            }
            for (int b = 0; b < nbits; b++)
            {
                int bit = (msg[b >> 3] >> (7 - (b & 7))) & 1;
                int at  = 16 + 2 * b + (bit ? 0 : 1);
                if (half) { level[at] += 0.5f; level[at + 1] += 0.5f; }
                else level[at] += 1.0f;
            }
        }
        else
        { /* half-microsecond slot k of the frame carries a pulse or not; slot k spans [k/2, (k+1)/2) us after the frame start, which
             lies `frac` of a sample into sample 0.  Each sample integrates what falls into its bin. */
            const double spu  = rate_x10 / 10.0;                                  /* samples per microsecond */
            const double frac = (half ? 0.5 : 0.0) + (double)sm64_below(&fs, 5) / 5.0 * 0.5; /* 0 .. 0.9 sample */
            nsamp             = (int)((8 + nbits) * spu + frac) + 2;
            for (int k = 0; k < (8 + nbits) * 2; k++)
            {
                int on;
                if (k < 16) on = (k == pre[0] || k == pre[1] || k == pre[2] || k == pre[3]);
                else
                {
                    int b   = (k - 16) >> 1;
                    int bit = (msg[b >> 3] >> (7 - (b & 7))) & 1;
                    on      = ((k & 1) == 0) ? bit : !bit;
                }
                if (!on) continue;
                const double t0 = frac + k * 0.5 * spu, t1 = t0 + 0.5 * spu; /* in samples */
                for (int sidx = (int)t0; sidx < 299 && (double)sidx < t1; sidx++)
                {
                    double lo = t0 > sidx ? t0 : sidx, hi = t1 < sidx + 1 ? t1 : sidx + 1;
                    if (hi > lo) level[sidx] += (float)(hi - lo);
                }
            }
        }
        for (int s = 0; s < nsamp + 1 && pos + (size_t)s < n; s++)
        {
            if (level[s] == 0.0f) continue;
            double a  = amp * (double)level[s];
            size_t o  = 2 * (pos + (size_t)s);
            out[o]     = clamp_u8((int)out[o] + (int)lround(a * cphi));
            out[o + 1] = clamp_u8((int)out[o + 1] + (int)lround(a * sphi));
        }
        if (frames && nframes < cap)
        {
            adsb_synth_frame_t* f = &frames[nframes];
            f->start              = (uint32_t)pos;
            f->nbits              = (uint8_t)nbits;
            f->flipped_bit        = (int8_t)flip;
            f->half_sample        = (uint8_t)half;
            f->amplitude          = (uint8_t)amp;
            memcpy(f->msg, msg, 14);
        }
        nframes++;
        pos += (size_t)nsamp + 16 + sm64_below(&fs, (uint32_t)gapmax);
    }
    return nframes;
}

int adsb_synth_fill(const adsb_synth_cfg_t* c, uint64_t buf_index, uint8_t* out, size_t nbytes, adsb_synth_frame_t* frames, int cap)
{
    return fill_impl(c, buf_index, out, nbytes, frames, cap, 20);
}
int adsb_synth_fill_rate(const adsb_synth_cfg_t* c, uint64_t buf_index, uint8_t* out, size_t nbytes, adsb_synth_frame_t* frames, int cap, int rate_x10)
{
    return fill_impl(c, buf_index, out, nbytes, frames, cap, rate_x10 == 24 ? 24 : 20);
}

typedef struct
{
    const adsb_synth_cfg_t* cfg;
    uint64_t                first, count;
    uint8_t*                out;
    size_t                  buf_bytes;
    int                     tid, nthreads;
    long                    frames;
    int                     rate_x10;
} fill_job_t;

static void* fill_worker(void* p)
{
    fill_job_t* j = (fill_job_t*)p;
    for (uint64_t b = (uint64_t)j->tid; b < j->count; b += (uint64_t)j->nthreads)
        j->frames += fill_impl(j->cfg, j->first + b, j->out + b * j->buf_bytes, j->buf_bytes, NULL, 0, j->rate_x10);
    return NULL;
}

static long fill_range_impl(const adsb_synth_cfg_t* c, uint64_t first_buf, uint64_t nbuf, uint8_t* out, size_t buf_bytes, int nthreads, int rate_x10);
long adsb_synth_fill_range(const adsb_synth_cfg_t* c, uint64_t first_buf, uint64_t nbuf, uint8_t* out, size_t buf_bytes, int nthreads)
{
    return fill_range_impl(c, first_buf, nbuf, out, buf_bytes, nthreads, 20);
}
long adsb_synth_fill_range_rate(const adsb_synth_cfg_t* c, uint64_t first_buf, uint64_t nbuf, uint8_t* out, size_t buf_bytes, int nthreads, int rate_x10)
{
    return fill_range_impl(c, first_buf, nbuf, out, buf_bytes, nthreads, rate_x10 == 24 ? 24 : 20);
}
static long fill_range_impl(const adsb_synth_cfg_t* c, uint64_t first_buf, uint64_t nbuf, uint8_t* out, size_t buf_bytes, int nthreads, int rate_x10)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    pthread_t  th[64];
    fill_job_t jobs[64];
    for (int t = 0; t < nthreads; t++)
    {
        jobs[t] = (fill_job_t){c, first_buf, nbuf, out, buf_bytes, t, nthreads, 0, rate_x10};
        pthread_create(&th[t], NULL, fill_worker, &jobs[t]);
    }
    long total = 0;
    for (int t = 0; t < nthreads; t++)
    {
        pthread_join(th[t], NULL);
        total += jobs[t].frames;
    }
    return total;
}
