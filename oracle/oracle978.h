/*
 * oracle978.h -- CPU restatement of the UAT 978 path (see oracle978.c for scope and pin status: parity unpinned).
 * TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 */
#ifndef ORACLE978_H
#define ORACLE978_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle978 oracle978_t;

/* dump_raw_message(updown, data, len, rs_errors) (reference uat2json-wrapper.cpp:14) plus the stream sample index of the sync word */
typedef void (*oracle978_cb)(void* user, char updown, const uint8_t* data, int len, int rs_errors, uint64_t sample_index);

oracle978_t* oracle978_create(void);
void         oracle978_destroy(oracle978_t* o);
/* UAT978Handler::HandleData (UAT978.cpp:43-60) */
void oracle978_handle_data(oracle978_t* o, const uint8_t* iq, size_t nbytes, oracle978_cb cb, void* user);
/* 0 (default): the reference's half-tail carry (UAT978.cpp:57, byte count = entry count); 1: carry the whole tail */
void     oracle978_set_carry_full(oracle978_t* o, int full);
uint64_t oracle978_offset(const oracle978_t* o);
size_t   oracle978_used(const oracle978_t* o);

/* the C seam of the reference (UAT978.cpp:9-10) */
void oracle978_init_fec(void);
int  oracle978_process_buffer(const uint16_t* phi, int len, uint64_t offset, oracle978_cb cb, void* user);

/* Reed-Solomon helpers: kind 0 = RS(30,18), 1 = RS(48,34), 2 = RS(92,72) */
void oracle978_rs_parity(int kind, const uint8_t* data, uint8_t* parity);
int  oracle978_rs_decode(int kind, uint8_t* codeword);

void oracle978_phase_lut(uint16_t* lut65536);

#ifdef __cplusplus
}
#endif
#endif
