/*
 * oracle978.c -- CPU restatement of the UAT 978 path behind libadsb's UAT978Handler.
 *
 * TEST INFRASTRUCTURE ONLY.  The product never calls this.
 *
 * What is restated from the reference tree itself:
 *   - the phase LUT                         UAT978.cpp:76-100   (oracle978_phase_lut, in oracle1090.c)
 *   - HandleData's LUT map + re-buffering   UAT978.cpp:43-60    (oracle978_handle_data)
 * What is NOT in the reference tree: process_buffer / init_fec and the Reed-Solomon decoder live in the un-vendored
 * third-party module github.com/ankurvdev/dump978 (fork of mutability/dump978, files legacy/dump978.c, legacy/fec.c,
 * libs/fec/{init,decode}_rs_char.c; pinned version unknown).  The functions below restate that
 * module's *published* algorithm (legacy dump978: two interleaved 18-bit sync searches on the sign of the phase
 * difference, re-check of the 36-bit sync word against a data-derived centre with <= 4 bit errors, slicing at that
 * centre, RS(30,18)/RS(48,34)/6xRS(92,72) over GF(256) poly 0x187, fcr 120, prim 1), from memory of the upstream source.
 *
 * PIN STATUS: parity unpinned.  No source, no capture and no usable golden exist for this part (the two 978 goldens
 * under tests/testdata are aircraft-level outputs of captures that are not in the snapshot).  The tests therefore
 * check self-consistency only: frames built by the generator, modulated, and recovered; GPU path == this file.
 */
#include "oracle978.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Reed-Solomon over GF(256), generator polynomial 0x187, first consecutive root 120, primitive element 1
 * (parameters named at reference CMakeLists.txt:92-93: libfec init_rs_char / decode_rs_char as dump978 uses them)
 * ---------------------------------------------------------------------------------------- */
#define NN 255
#define A0 255

typedef struct
{
    uint8_t alpha_to[256], index_of[256];
    uint8_t genpoly[33]; /* index form */
    int     nroots, fcr, prim, iprim, pad;
} rs_t;

static int modnn(int x)
{
    while (x >= NN)
    {
        x -= NN;
        x = (x >> 8) + (x & NN);
    }
    return x;
}

static void rs_init(rs_t* rs, int gfpoly, int fcr, int prim, int nroots, int pad)
{
    memset(rs, 0, sizeof(*rs));
    rs->nroots = nroots;
    rs->fcr    = fcr;
    rs->prim   = prim;
    rs->pad    = pad;
    rs->index_of[0]  = A0;
    rs->alpha_to[A0] = 0;
    int sr = 1;
    for (int i = 0; i < NN; i++)
    {
        rs->index_of[sr] = (uint8_t)i;
        rs->alpha_to[i]  = (uint8_t)sr;
        sr <<= 1;
        if (sr & 0x100) sr ^= gfpoly;
        sr &= NN;
    }
    int iprim;
    for (iprim = 1; (iprim % prim) != 0; iprim += NN) {}
    rs->iprim = iprim / prim;
    uint8_t g[33];
    memset(g, 0, sizeof(g));
    g[0] = 1;
    for (int i = 0, root = fcr * prim; i < nroots; i++, root += prim)
    {
        g[i + 1] = 1;
        for (int j = i; j > 0; j--)
        {
            if (g[j] != 0) g[j] = g[j - 1] ^ rs->alpha_to[modnn(rs->index_of[g[j]] + root)];
            else g[j] = g[j - 1];
        }
        g[0] = rs->alpha_to[modnn(rs->index_of[g[0]] + root)];
    }
    for (int i = 0; i <= nroots; i++) rs->genpoly[i] = rs->index_of[g[i]];
}

static void rs_encode(const rs_t* rs, const uint8_t* data, uint8_t* parity)
{
    const int nr = rs->nroots, k = NN - nr - rs->pad;
    memset(parity, 0, (size_t)nr);
    for (int i = 0; i < k; i++)
    {
        int fb = rs->index_of[data[i] ^ parity[0]];
        if (fb != A0)
            for (int j = 1; j < nr; j++) parity[j] ^= rs->alpha_to[modnn(fb + rs->genpoly[nr - j])];
        memmove(parity, parity + 1, (size_t)nr - 1);
        parity[nr - 1] = (fb != A0) ? rs->alpha_to[modnn(fb + rs->genpoly[0])] : 0;
    }
}

/* In-place decode of one (255-pad)-byte codeword; returns the number of corrected symbols or -1. */
static int rs_decode(const rs_t* rs, uint8_t* data)
{
    const int nr = rs->nroots, n = NN - rs->pad;
    uint8_t   s[32], lambda[33], b[33], t[33], omega[33], root[32], reg[33], loc[32];
    int       syn_error = 0;

    for (int i = 0; i < nr; i++) s[i] = data[0];
    for (int j = 1; j < n; j++)
        for (int i = 0; i < nr; i++)
        {
            if (s[i] == 0) s[i] = data[j];
            else s[i] = data[j] ^ rs->alpha_to[modnn(rs->index_of[s[i]] + (rs->fcr + i) * rs->prim)];
        }
    for (int i = 0; i < nr; i++)
    {
        syn_error |= s[i];
        s[i] = rs->index_of[s[i]];
    }
    if (!syn_error) return 0;

    memset(lambda + 1, 0, (size_t)nr);
    lambda[0] = 1;
    for (int i = 0; i < nr + 1; i++) b[i] = rs->index_of[lambda[i]];

    /* Berlekamp-Massey */
    int r = 0, el = 0;
    while (++r <= nr)
    {
        uint8_t discr = 0;
        for (int i = 0; i < r; i++)
            if (lambda[i] != 0 && s[r - i - 1] != A0) discr ^= rs->alpha_to[modnn(rs->index_of[lambda[i]] + s[r - i - 1])];
        discr = rs->index_of[discr];
        if (discr == A0)
        {
            memmove(b + 1, b, (size_t)nr);
            b[0] = A0;
        }
        else
        {
            t[0] = lambda[0];
            for (int i = 0; i < nr; i++)
            {
                if (b[i] != A0) t[i + 1] = lambda[i + 1] ^ rs->alpha_to[modnn(discr + b[i])];
                else t[i + 1] = lambda[i + 1];
            }
            if (2 * el <= r - 1)
            {
                el = r - el;
                for (int i = 0; i <= nr; i++) b[i] = (lambda[i] == 0) ? A0 : (uint8_t)modnn(rs->index_of[lambda[i]] - discr + NN);
            }
            else
            {
                memmove(b + 1, b, (size_t)nr);
                b[0] = A0;
            }
            memcpy(lambda, t, (size_t)nr + 1);
        }
    }

    int deg_lambda = 0;
    for (int i = 0; i < nr + 1; i++)
    {
        lambda[i] = rs->index_of[lambda[i]];
        if (lambda[i] != A0) deg_lambda = i;
    }
    /* Chien search */
    memcpy(reg + 1, lambda + 1, (size_t)nr);
    int count = 0;
    for (int i = 1, k = rs->iprim - 1; i <= NN; i++, k = modnn(k + rs->iprim))
    {
        uint8_t q = 1;
        for (int j = deg_lambda; j > 0; j--)
            if (reg[j] != A0)
            {
                reg[j] = (uint8_t)modnn(reg[j] + j);
                q ^= rs->alpha_to[reg[j]];
            }
        if (q != 0) continue;
        root[count] = (uint8_t)i;
        loc[count]  = (uint8_t)k;
        if (++count == deg_lambda) break;
    }
    if (deg_lambda != count) return -1;

    /* omega(x) = s(x) * lambda(x) mod x^nroots, index form */
    int deg_omega = deg_lambda - 1;
    for (int i = 0; i <= deg_omega; i++)
    {
        uint8_t tmp = 0;
        for (int j = i; j >= 0; j--)
            if (s[i - j] != A0 && lambda[j] != A0) tmp ^= rs->alpha_to[modnn(s[i - j] + lambda[j])];
        omega[i] = rs->index_of[tmp];
    }
    /* Forney */
    for (int j = count - 1; j >= 0; j--)
    {
        uint8_t num1 = 0;
        for (int i = deg_omega; i >= 0; i--)
            if (omega[i] != A0) num1 ^= rs->alpha_to[modnn(omega[i] + i * root[j])];
        uint8_t num2 = rs->alpha_to[modnn(root[j] * (rs->fcr - 1) + NN)];
        uint8_t den  = 0;
        int     lim  = (deg_lambda < nr - 1 ? deg_lambda : nr - 1) & ~1;
        for (int i = lim; i >= 0; i -= 2)
            if (lambda[i + 1] != A0) den ^= rs->alpha_to[modnn(lambda[i + 1] + i * root[j])];
        /* libfec (non-DEBUG build) neither rejects den == 0 (index_of[0] = 255 simply enters the exponent) nor an error
         * located in the virtual padding: that correction is dropped, the count still includes it */
        if (num1 != 0 && loc[j] >= rs->pad)
            data[loc[j] - rs->pad] ^= rs->alpha_to[modnn(rs->index_of[num1] + rs->index_of[num2] + NN - rs->index_of[den])];
    }
    return count;
}

static rs_t g_rs_short, g_rs_long, g_rs_uplink;
static int  g_rs_ready = 0;

void oracle978_init_fec(void) /* dump978 init_fec(): the three codes UAT uses */
{
    if (g_rs_ready) return;
    rs_init(&g_rs_short, 0x187, 120, 1, 12, 225);  /* RS(30,18)  */
    rs_init(&g_rs_long, 0x187, 120, 1, 14, 207);   /* RS(48,34)  */
    rs_init(&g_rs_uplink, 0x187, 120, 1, 20, 163); /* RS(92,72)  */
    g_rs_ready = 1;
}

void oracle978_rs_parity(int kind, const uint8_t* data, uint8_t* parity)
{
    oracle978_init_fec();
    rs_encode(kind == 0 ? &g_rs_short : kind == 1 ? &g_rs_long : &g_rs_uplink, data, parity);
}
int oracle978_rs_decode(int kind, uint8_t* codeword)
{
    oracle978_init_fec();
    return rs_decode(kind == 0 ? &g_rs_short : kind == 1 ? &g_rs_long : &g_rs_uplink, codeword);
}

/* ------------------------------------------------------------------------------------------
 * frame demodulation (dump978 legacy: check_sync_word, demod_frame, correct_*_frame, process_buffer)
 * ---------------------------------------------------------------------------------------- */
#define SYNC_BITS 36
#define CHECK_BITS 18
#define MAX_SYNC_ERRORS 4
#define ADSB_SYNC_WORD 0xEACDDA4E2ULL
#define UPLINK_SYNC_WORD 0x153225B1DULL
#define SHORT_DATA_BYTES 18
#define SHORT_BYTES 30
#define LONG_DATA_BYTES 34
#define LONG_BYTES 48
#define LONG_BITS (LONG_BYTES * 8)
#define SHORT_BITS (SHORT_BYTES * 8)
#define UPLINK_BLOCK_DATA 72
#define UPLINK_BLOCK_BYTES 92
#define UPLINK_BLOCKS 6
#define UPLINK_BYTES (UPLINK_BLOCK_BYTES * UPLINK_BLOCKS) /* 552 */
#define UPLINK_BITS (UPLINK_BYTES * 8)                    /* 4416 */
#define UPLINK_DATA_BYTES (UPLINK_BLOCK_DATA * UPLINK_BLOCKS)

static inline int16_t phi_difference(uint16_t from, uint16_t to)
{
    int32_t d = (int32_t)to - (int32_t)from; /* wrap into [-32768, 32767] */
    if (d >= 32768) return (int16_t)(d - 65536);
    if (d < -32768) return (int16_t)(d + 65536);
    return (int16_t)d;
}

/* mean dphi of the zero bits and of the one bits of the sync word -> centre; then count bits on the wrong side */
static int check_sync_word(const uint16_t* phi, uint64_t pattern, int16_t* center)
{
    int32_t zero_total = 0, one_total = 0;
    int     zero_bits = 0, one_bits = 0;
    for (int i = 0; i < SYNC_BITS; i++)
    {
        int16_t d = phi_difference(phi[i * 2], phi[i * 2 + 1]);
        if (pattern & (1ULL << (35 - i)))
        {
            one_bits++;
            one_total += d;
        }
        else
        {
            zero_bits++;
            zero_total += d;
        }
    }
    zero_total /= zero_bits;
    one_total /= one_bits;
    *center = (int16_t)((one_total + zero_total) / 2);
    int errors = 0;
    for (int i = 0; i < SYNC_BITS; i++)
    {
        int16_t d = phi_difference(phi[i * 2], phi[i * 2 + 1]);
        if (pattern & (1ULL << (35 - i)))
        {
            if (d < *center) errors++;
        }
        else if (d > *center) errors++;
    }
    return errors <= MAX_SYNC_ERRORS;
}

static void demod_frame(const uint16_t* phi, uint8_t* to, int bytes, int16_t center)
{
    for (int k = 0; k < bytes; k++)
    {
        uint8_t b = 0;
        for (int i = 0; i < 8; i++) b = (uint8_t)((b << 1) | (phi_difference(phi[(k * 8 + i) * 2], phi[(k * 8 + i) * 2 + 1]) > center ? 1 : 0));
        to[k] = b;
    }
}

/* 2: long frame, 1: short frame, 0: neither */
static int correct_adsb_frame(uint8_t* to, int* rs_errors)
{
    /* in place, as upstream: a long decode that succeeds on a frame whose type field is 0 leaves its corrections in the
     * buffer the short decode then sees (an uncorrectable decode leaves the data alone) */
    int n = rs_decode(&g_rs_long, to);
    if (n >= 0 && n <= 7 && (to[0] >> 3) != 0)
    {
        *rs_errors = n;
        return 2;
    }
    n = rs_decode(&g_rs_short, to);
    if (n >= 0 && n <= 6 && (to[0] >> 3) == 0)
    {
        *rs_errors = n;
        return 1;
    }
    *rs_errors = 9999;
    return 0;
}

static int demod_adsb_frame(const uint16_t* phi, uint8_t* to, int* rs_errors)
{
    int16_t center;
    if (!check_sync_word(phi, ADSB_SYNC_WORD, &center))
    {
        *rs_errors = 9999;
        return 0;
    }
    demod_frame(phi + SYNC_BITS * 2, to, LONG_BYTES, center);
    int type = correct_adsb_frame(to, rs_errors);
    if (type == 2) return SYNC_BITS + LONG_BITS;
    if (type == 1) return SYNC_BITS + SHORT_BITS;
    return 0;
}

static int correct_uplink_frame(const uint8_t* from, uint8_t* to, int* rs_errors)
{
    int total = 0;
    for (int block = 0; block < UPLINK_BLOCKS; block++)
    {
        uint8_t cw[UPLINK_BLOCK_BYTES];
        for (int i = 0; i < UPLINK_BLOCK_BYTES; i++) cw[i] = from[i * UPLINK_BLOCKS + block]; /* de-interleave */
        int n = rs_decode(&g_rs_uplink, cw);
        if (n < 0 || n > 10)
        {
            *rs_errors = 9999;
            return 0;
        }
        total += n;
        memcpy(to + block * UPLINK_BLOCK_DATA, cw, UPLINK_BLOCK_DATA);
    }
    *rs_errors = total;
    return 1;
}

static int demod_uplink_frame(const uint16_t* phi, uint8_t* to, int* rs_errors)
{
    int16_t center;
    uint8_t raw[UPLINK_BYTES];
    if (!check_sync_word(phi, UPLINK_SYNC_WORD, &center))
    {
        *rs_errors = 9999;
        return 0;
    }
    demod_frame(phi + SYNC_BITS * 2, raw, UPLINK_BYTES, center);
    if (!correct_uplink_frame(raw, to, rs_errors)) return 0;
    return SYNC_BITS + UPLINK_BITS;
}

int oracle978_process_buffer(const uint16_t* phi, int len, uint64_t offset, oracle978_cb cb, void* user)
{
    oracle978_init_fec();
    const uint64_t check_mask   = (1ULL << CHECK_BITS) - 1;
    const uint64_t check_adsb   = ADSB_SYNC_WORD >> (SYNC_BITS - CHECK_BITS);
    const uint64_t check_uplink = UPLINK_SYNC_WORD >> (SYNC_BITS - CHECK_BITS);
    uint64_t       sync0 = 0, sync1 = 0;
    uint8_t        buf_a[UPLINK_DATA_BYTES > LONG_BYTES ? UPLINK_DATA_BYTES : LONG_BYTES], buf_b[sizeof(buf_a)];
    /* stop while a maximum-size frame still fits; the caller passes the tail again, so no state is kept */
    const int lenbits = len / 2 - (SYNC_BITS + UPLINK_BITS);
    int       bit;
    for (bit = 0; bit < lenbits; bit++)
    {
        int16_t d0 = phi_difference(phi[bit * 2], phi[bit * 2 + 1]);
        int16_t d1 = phi_difference(phi[bit * 2 + 1], phi[bit * 2 + 2]);
        sync0      = ((sync0 << 1) | (d0 > 0 ? 1 : 0)) & check_mask;
        sync1      = ((sync1 << 1) | (d1 > 0 ? 1 : 0)) & check_mask;
        if (bit < CHECK_BITS) continue;
        if (sync0 == check_adsb || sync1 == check_adsb)
        {
            int startbit = bit - CHECK_BITS + 1;
            int shift    = (sync0 == check_adsb) ? 0 : 1;
            int index    = startbit * 2 + shift;
            int rs0 = -1, rs1 = -1;
            int skip0 = demod_adsb_frame(phi + index, buf_a, &rs0);
            int skip1 = demod_adsb_frame(phi + index + 1, buf_b, &rs1);
            if (skip0 && rs0 <= rs1)
            {
                if (cb) cb(user, '-', buf_a, (buf_a[0] >> 3) == 0 ? SHORT_DATA_BYTES : LONG_DATA_BYTES, rs0, offset + (uint64_t)index);
                bit = startbit + skip0;
                continue;
            }
            else if (skip1 && rs1 <= rs0)
            {
                if (cb) cb(user, '-', buf_b, (buf_b[0] >> 3) == 0 ? SHORT_DATA_BYTES : LONG_DATA_BYTES, rs1, offset + (uint64_t)index + 1);
                bit = startbit + skip1;
                continue;
            }
        }
        else if (sync0 == check_uplink || sync1 == check_uplink) /* only reached when neither register held the ADS-B word */
        {
            int startbit = bit - CHECK_BITS + 1;
            int shift    = (sync0 == check_uplink) ? 0 : 1;
            int index    = startbit * 2 + shift;
            int rs0 = -1, rs1 = -1;
            int skip0 = demod_uplink_frame(phi + index, buf_a, &rs0);
            int skip1 = demod_uplink_frame(phi + index + 1, buf_b, &rs1);
            if (skip0 && rs0 <= rs1)
            {
                if (cb) cb(user, '+', buf_a, UPLINK_DATA_BYTES, rs0, offset + (uint64_t)index);
                bit = startbit + skip0;
                continue;
            }
            else if (skip1 && rs1 <= rs0)
            {
                if (cb) cb(user, '+', buf_b, UPLINK_DATA_BYTES, rs1, offset + (uint64_t)index + 1);
                bit = startbit + skip1;
                continue;
            }
        }
    }
    return (bit - CHECK_BITS) * 2;
}

/* ------------------------------------------------------------------------------------------
 * UAT978Handler::HandleData (UAT978.cpp:43-60): LUT map into a 65536-entry staging buffer, process, keep the tail
 * ---------------------------------------------------------------------------------------- */
struct oracle978
{
    uint16_t lut[65536];
    uint16_t buffer[65536];
    size_t   used;
    uint64_t offset;
    int      carry_full;
};

extern void oracle978_phase_lut(uint16_t* lut65536); /* oracle1090.c */

oracle978_t* oracle978_create(void)
{
    oracle978_t* o = (oracle978_t*)calloc(1, sizeof(*o));
    oracle978_phase_lut(o->lut);
    oracle978_init_fec();
    return o;
}
void oracle978_destroy(oracle978_t* o) { free(o); }
void oracle978_set_carry_full(oracle978_t* o, int full) { o->carry_full = full; }
uint64_t oracle978_offset(const oracle978_t* o) { return o->offset; }
size_t   oracle978_used(const oracle978_t* o) { return o->used; }

void oracle978_handle_data(oracle978_t* o, const uint8_t* iq, size_t nbytes, oracle978_cb cb, void* user)
{
    const size_t n = nbytes / 2;
    size_t       j = 0;
    while (j < n)
    {
        size_t i = o->used;
        for (; i < 65536 && j < n; i++, j++) o->buffer[i] = o->lut[(uint32_t)iq[2 * j] | ((uint32_t)iq[2 * j + 1] << 8)];
        int done = oracle978_process_buffer(o->buffer, (int)i, o->offset, cb, user);
        /* Fewer than 2 * (36 + 4416) + 2 staged samples: process_buffer returns -36 and UAT978.cpp:55-58 would move the
         * offset backwards and memmove from in front of the array.  That is outside anything restatable; production calls
         * (262 144 B) never get there.  Oracle and product both keep the staged data and consume nothing. */
        if (done < 0) done = 0;
        o->offset += (uint64_t)done;
        /* UAT978.cpp:57 passes the tail length in ENTRIES as memmove's BYTE count, so only the first half of the unconsumed
         * tail moves to the front; the second half of the "carried" region keeps whatever the staging buffer held there
         * (phases of samples `done` earlier in the stream).  `used` (:58) still counts the whole tail.  Restated as is;
         * carry_full = 1 selects the evidently intended behaviour instead. */
        const size_t tail = i - (size_t)done;
        memmove(o->buffer, o->buffer + done, o->carry_full ? tail * sizeof(uint16_t) : tail);
        o->used = tail;
    }
}
