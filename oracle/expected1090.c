/*
 * expected1090.c -- the record set the GPU scan must produce, built from the oracle's state-free probes.
 *
 * TEST INFRASTRUCTURE ONLY (part of liboracle1090.so; see oracle1090.h).  For every offset that passes both preamble
 * gates of the restated reference (oracle1090_gate_offsets, ADSB1090.cpp:782-811) it applies the emission contract of
 * include/adsb_amd.h (adsb_amd_record_t) to what oracle1090_probe_at reports for the two slicing passes
 * (ADSB1090.cpp:814-881, 491-529): one record per (offset, pass) that the reference could accept.  The same rules exist
 * in Python (tests/helpers.py: records_from_probe) for small inputs; this is the version that covers a whole 1 GiB input
 * in seconds (buffers spread over threads).
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "oracle1090.h"

/* mirrors adsb_amd_record_t (include/adsb_amd.h): 32 bytes, little endian */
typedef struct expected_record
{
    uint32_t buffer, offset, addr;
    uint16_t reserved;
    uint8_t  nbits;
    int8_t   errorbit;
    uint8_t  df, flags;
    uint8_t  msg[14];
} expected_record_t;

enum { F_PASS2 = 1, F_PHASE = 2, F_NEEDS_ICAO = 4 };

static int is_ap(int df) { return df == 0 || df == 4 || df == 5 || df == 16 || df == 20 || df == 21 || df == 24; }

static void put(expected_record_t* r, uint32_t buffer, uint32_t j, const uint8_t* msg, int nbits, int errorbit, int df, int flags, uint32_t addr)
{
    memset(r, 0, sizeof(*r));
    r->buffer   = buffer;
    r->offset   = j;
    r->addr     = addr;
    r->nbits    = (uint8_t)nbits;
    r->errorbit = (int8_t)errorbit;
    r->df       = (uint8_t)df;
    r->flags    = (uint8_t)flags;
    memcpy(r->msg, msg, (size_t)(nbits / 8)); /* contract: bytes beyond the message length are zero */
}

/* records of one offset; returns how many (0..2) */
static int records_of_probe(uint32_t buffer, uint32_t j, const oracle1090_probe_t* p, expected_record_t out[2])
{
    int n = 0;
    if (p->p_errors[0] != 0) return 0; /* bit 0 is never rescaled: the retry cannot clear it */
    if (!p->p_energy_ok[0]) return 0;
    for (int k = 0; k < 2; k++)
    {
        if (k == 1 && (!p->phase_applied || !p->p_energy_ok[1] || p->p_errors[1] != 0)) break;
        const int df = p->p_df[k], nbits = p->p_nbits[k];
        const int flags = k == 0 ? 0 : (F_PASS2 | F_PHASE);
        if (df == 11 || df == 17)
        {
            if (p->p_crc_state[k] == 1 || p->p_crc_state[k] == 2)
            {
                const uint8_t* m = p->p_fixed[k];
                put(&out[n++], buffer, j, m, nbits, p->p_errorbit[k], df, flags, ((uint32_t)m[1] << 16) | ((uint32_t)m[2] << 8) | m[3]);
                break; /* accepted without state: the reference never retries */
            }
        }
        else if (is_ap(df)) put(&out[n++], buffer, j, p->p_msg[k], nbits, -1, df, flags | F_NEEDS_ICAO, p->p_ap_addr[k]);
    }
    return n;
}

typedef struct job
{
    const uint8_t*     iq;
    size_t             buffer_bytes, nbuf;
    int                tid, nthreads;
    expected_record_t* out; /* per-thread private array */
    size_t             count, cap;
    size_t*            per_buffer; /* shared: records of buffer b */
    int                failed;
} job_t;

static void* worker(void* arg)
{
    job_t*    jb  = (job_t*)arg;
    size_t    n   = jb->buffer_bytes / 2;
    uint16_t* mag = (uint16_t*)malloc(n * sizeof(uint16_t) + 16);
    size_t    gcap = n / 8 + 1024;
    uint32_t* gate = (uint32_t*)malloc(gcap * sizeof(uint32_t));
    if (!mag || !gate) { jb->failed = 1; free(mag); free(gate); return NULL; }
    for (size_t b = (size_t)jb->tid; b < jb->nbuf; b += (size_t)jb->nthreads)
    {
        oracle1090_magnitude(jb->iq + b * jb->buffer_bytes, jb->buffer_bytes, mag);
        size_t ng = oracle1090_gate_offsets(mag, n, gate, gcap);
        if (ng > gcap)
        {
            gcap = ng;
            free(gate);
            gate = (uint32_t*)malloc(gcap * sizeof(uint32_t));
            if (!gate) { jb->failed = 1; break; }
            ng = oracle1090_gate_offsets(mag, n, gate, gcap);
        }
        size_t before = jb->count;
        for (size_t g = 0; g < ng; g++)
        {
            oracle1090_probe_t p;
            oracle1090_probe_at(mag, n, gate[g], &p);
            if (jb->count + 2 > jb->cap)
            {
                jb->cap = jb->cap * 2 + 1024;
                jb->out = (expected_record_t*)realloc(jb->out, jb->cap * sizeof(expected_record_t));
                if (!jb->out) { jb->failed = 1; break; }
            }
            jb->count += (size_t)records_of_probe((uint32_t)b, gate[g], &p, jb->out + jb->count);
        }
        if (jb->failed) break;
        jb->per_buffer[b] = jb->count - before;
    }
    free(mag);
    free(gate);
    return NULL;
}

/* Expected record array of one scan call over `iq` split into buffers of `buffer_bytes` (0: one buffer = the whole input, inputs
 * under 480 bytes give nothing; a trailing partial buffer is ignored).  Writes up to `cap` records to `out` in (buffer, offset,
 * pass) order and returns the total count (may exceed cap: call again), or (size_t)-1 on allocation failure. */
size_t oracle1090_expected_records(const uint8_t* iq, size_t nbytes, size_t buffer_bytes, void* out, size_t cap, int nthreads)
{
    size_t bb   = buffer_bytes ? buffer_bytes : (nbytes & ~(size_t)1);
    size_t nbuf = bb ? nbytes / bb : 0;
    if (buffer_bytes == 0 && nbytes < 480) nbuf = 0;
    if (nbuf == 0) return 0;
    if (bb / 2 <= 240) return 0; /* no position j < N - 240 exists (the reference's loop bound, :772) */
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > nbuf) nthreads = (int)nbuf;
    (void)oracle1090_mag_lut(); /* tables are built lazily: do it before the threads start */
    (void)oracle1090_checksum_entry(0);
    job_t*     jobs = (job_t*)calloc((size_t)nthreads, sizeof(job_t));
    pthread_t* th   = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    size_t*    per  = (size_t*)calloc(nbuf, sizeof(size_t));
    if (!jobs || !th || !per) { free(jobs); free(th); free(per); return (size_t)-1; }
    for (int t = 0; t < nthreads; t++)
    {
        jobs[t].iq = iq; jobs[t].buffer_bytes = bb; jobs[t].nbuf = nbuf; jobs[t].tid = t; jobs[t].nthreads = nthreads; jobs[t].per_buffer = per;
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    int failed = 0;
    for (int t = 0; t < nthreads; t++)
    {
        pthread_join(th[t], NULL);
        failed |= jobs[t].failed;
    }
    size_t total = 0;
    if (!failed)
    {
        /* stitch the per-thread arrays back into buffer order: thread t produced buffers t, t+T, ... in that order */
        size_t*            cursor = (size_t*)calloc((size_t)nthreads, sizeof(size_t));
        expected_record_t* dst    = (expected_record_t*)out;
        for (size_t b = 0; b < nbuf && cursor; b++)
        {
            job_t* jb = &jobs[b % (size_t)nthreads];
            for (size_t k = 0; k < per[b]; k++, total++)
                if (total < cap) dst[total] = jb->out[cursor[b % (size_t)nthreads] + k];
            cursor[b % (size_t)nthreads] += per[b];
        }
        if (!cursor) failed = 1;
        free(cursor);
    }
    for (int t = 0; t < nthreads; t++) free(jobs[t].out);
    free(jobs);
    free(th);
    free(per);
    return failed ? (size_t)-1 : total;
}
