/* synth1090.h -- deterministic synthetic 1090ES u8 IQ generator (see synth1090.c). */
#ifndef ADSB_AMD_SYNTH1090_H
#define ADSB_AMD_SYNTH1090_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct adsb_synth_cfg
{
    uint64_t seed;           /* 0x1090AD5B */
    int32_t  noise_amp;      /* background: uniform integer in [-noise_amp, +noise_amp] around 127 */
    int32_t  mean_spacing;   /* mean frame start-to-start spacing in samples; <=0 disables frames */
    int32_t  amp_lo, amp_hi; /* pulse amplitude range in LSB */
    int32_t  pct_df17, pct_df11; /* remainder: DF4/5/20/21 with AP = CRC xor ICAO */
    int32_t  pct_bitflip;    /* frames with exactly one flipped bit */
    int32_t  pct_halfsample; /* frames generated with a half-sample timing offset */
    int32_t  pool_size;      /* ICAO address pool */
} adsb_synth_cfg_t;

typedef struct adsb_synth_frame
{
    uint32_t start; /* sample index of the first preamble pulse inside the buffer */
    uint8_t  msg[14];
    uint8_t  nbits;
    int8_t   flipped_bit; /* -1: none */
    uint8_t  half_sample;
    uint8_t  amplitude;
} adsb_synth_frame_t;

void     adsb_synth_default(adsb_synth_cfg_t* cfg);
uint32_t adsb_synth_pool_addr(const adsb_synth_cfg_t* cfg, uint32_t k);
/* Fill one buffer (stream index buf_index).  Returns the number of frames injected. */
int  adsb_synth_fill(const adsb_synth_cfg_t* cfg, uint64_t buf_index, uint8_t* out, size_t nbytes, adsb_synth_frame_t* frames, int cap);
/* Fill nbuf consecutive buffers with nthreads worker threads.  Returns total frames injected. */
long adsb_synth_fill_range(const adsb_synth_cfg_t* cfg, uint64_t first_buf, uint64_t nbuf, uint8_t* out, size_t buf_bytes, int nthreads);

/* The same with a sample rate: rate_x10 = 20 (2.0 MS/s, identical bytes to the functions above) or 24 (2.4 MS/s: the pulse train
 * integrated over 1/2.4 us bins, frames at random sub-sample offsets; a throughput workload, nothing in the reference decodes it). */
int  adsb_synth_fill_rate(const adsb_synth_cfg_t* cfg, uint64_t buf_index, uint8_t* out, size_t nbytes, adsb_synth_frame_t* frames, int cap, int rate_x10);
long adsb_synth_fill_range_rate(const adsb_synth_cfg_t* cfg, uint64_t first_buf, uint64_t nbuf, uint8_t* out, size_t buf_bytes, int nthreads, int rate_x10);

#ifdef __cplusplus
}
#endif
#endif
