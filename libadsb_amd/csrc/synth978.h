/* synth978.h -- deterministic synthetic UAT 978 u8 IQ generator (see synth978.c). */
#ifndef ADSB_AMD_SYNTH978_H
#define ADSB_AMD_SYNTH978_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct adsb_synth978_cfg
{
    uint64_t seed;
    int32_t  noise_amp;      /* uniform integer noise on I and Q, +-noise_amp */
    int32_t  amp_lo, amp_hi; /* carrier amplitude range in LSB */
    int32_t  mean_gap_bits;  /* mean idle time between frames, in bit periods */
    int32_t  pct_uplink;     /* share of ground uplink frames */
    int32_t  pct_long;       /* share of long frames among downlink frames */
    int32_t  pct_corrupt;    /* share of frames with corrupted code-word bytes */
    int32_t  max_bad_bytes;  /* 1..max_bad_bytes corrupted bytes in such a frame */
} adsb_synth978_cfg_t;

typedef struct adsb_synth978_frame
{
    uint64_t start; /* sample index of the first sync bit */
    uint8_t  kind;  /* 0 short downlink, 1 long downlink, 2 uplink */
    uint8_t  bad_bytes;
    uint16_t len;   /* payload bytes: 18 / 34 / 432 */
    uint8_t  data[432];
} adsb_synth978_frame_t;

void adsb_synth978_default(adsb_synth978_cfg_t* cfg);
long adsb_synth978_fill(const adsb_synth978_cfg_t* cfg, uint64_t stream_index, uint8_t* out, size_t nbytes, adsb_synth978_frame_t* frames, long cap);
#ifdef __cplusplus
}
#endif
#endif
