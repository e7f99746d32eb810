/*
 * pq_oracle.h -- CPU restatement of parquet-go's page-value decoders.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker the GPU decode path
 * is compared against (tests/, the smoke check and the CPU-baseline leg of
 * the benchmark).  Nothing in the product library may call, link or load it.
 *
 * Each function restates, in scalar C, the semantics of one decoder of the
 * reference (the purego variants are normative); the citation sits on the
 * function.  Encoders are included only to build test inputs shaped like the
 * reference writer's output.
 *
 * Parity status: PINNED against the reference's own known-answer vectors,
 * input tables and parquet test files (see tests/test_oracle_*.py); the one
 * external dependency (github.com/parquet-go/bitpack v1.0.3, Unpack/Pack) is
 * restated from the bit layout the reference's tests spell out
 * (encoding/delta/binary_packed_test.go:113-143,293-322).
 */
#ifndef PQ_ORACLE_H
#define PQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/pqd_status.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pqo_result {
    int32_t  status;    /* pqd_status */
    int64_t  arg;       /* status argument */
    uint64_t n_values;  /* values (or offsets-1) produced */
    uint64_t n_bytes;   /* bytes written to dst */
    uint64_t consumed;  /* source bytes consumed (DELTA streams) */
} pqo_result;

/* ---- decoders -------------------------------------------------------- */
pqo_result pqo_rle_decode_levels(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, uint32_t bit_width);
pqo_result pqo_rle_decode_boolean(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n);
pqo_result pqo_rle_decode_int32(int32_t* dst, uint64_t cap_bytes, const uint8_t* src, uint64_t n, uint32_t bit_width);
pqo_result pqo_rle_dict_decode_int32(int32_t* dst, uint64_t cap_bytes, const uint8_t* src, uint64_t n);
pqo_result pqo_bitpacked_decode_levels(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, uint32_t bit_width);
pqo_result pqo_delta_decode_int32(int32_t* dst, uint64_t cap_bytes, const uint8_t* src, uint64_t n);
pqo_result pqo_delta_decode_int64(int64_t* dst, uint64_t cap_bytes, const uint8_t* src, uint64_t n);
pqo_result pqo_dlba_decode(uint8_t* dst, uint64_t cap, uint32_t* offsets, uint64_t offsets_cap,
                           const uint8_t* src, uint64_t n);
pqo_result pqo_dba_decode(uint8_t* dst, uint64_t cap, uint32_t* offsets, uint64_t offsets_cap,
                          const uint8_t* src, uint64_t n);
pqo_result pqo_dba_decode_flba(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, int64_t size);
pqo_result pqo_bss_decode(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, int64_t size, int is_flba);
pqo_result pqo_plain_decode_fixed(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, int64_t width, int is_flba);
pqo_result pqo_plain_decode_byte_array(uint8_t* dst, uint64_t cap, uint32_t* offsets, uint64_t offsets_cap,
                                       const uint8_t* src, uint64_t n);
int64_t pqo_dlba_estimate(const uint8_t* src, uint64_t n);
int64_t pqo_dba_estimate(const uint8_t* src, uint64_t n);

/* dictionary surface */
pqo_result pqo_dict_lookup(uint8_t* rows, uint64_t row_stride, const uint8_t* dict, uint64_t dict_len,
                           uint32_t width, const int32_t* indexes, uint64_t n);
void pqo_indexed_page_fixup(int32_t* values, uint64_t decoded, uint64_t num_values);

/* ---- encoders (fixture generation only) ------------------------------ */
uint64_t pqo_rle_encode_levels(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, uint32_t bit_width);
uint64_t pqo_rle_encode_int32(uint8_t* dst, uint64_t cap, const int32_t* src, uint64_t n, uint32_t bit_width);
uint64_t pqo_rle_encode_boolean(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n);
uint64_t pqo_rle_dict_encode_int32(uint8_t* dst, uint64_t cap, const int32_t* src, uint64_t n);
uint64_t pqo_bitpacked_encode_levels(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, uint32_t bit_width);
uint64_t pqo_delta_encode_int32(uint8_t* dst, uint64_t cap, const int32_t* src, uint64_t n,
                                uint32_t block_size, uint32_t num_miniblocks);
uint64_t pqo_delta_encode_int64(uint8_t* dst, uint64_t cap, const int64_t* src, uint64_t n,
                                uint32_t block_size, uint32_t num_miniblocks);
uint64_t pqo_dlba_encode(uint8_t* dst, uint64_t cap, const uint8_t* src, const uint32_t* offsets, uint64_t n_offsets);
uint64_t pqo_dba_encode(uint8_t* dst, uint64_t cap, const uint8_t* src, const uint32_t* offsets, uint64_t n_offsets);
uint64_t pqo_bss_encode(uint8_t* dst, const uint8_t* src, uint64_t n, uint64_t size);
uint64_t pqo_plain_encode_byte_array(uint8_t* dst, uint64_t cap, const uint8_t* src, const uint32_t* offsets, uint64_t n_offsets);

int pqo_selftest_unpack(void);

/* ---- tuned ports of the benchmark's two decoders (pq_port_opt.c; cpu_baseline "port-opt") ---- */
pqo_result pqopt_delta_decode_int64(int64_t* dst, uint64_t cap_bytes, const uint8_t* src, uint64_t n);
/* values of an RLE_DICTIONARY int64 page: indexes decoded, fixed up to num_values, looked up (dst holds num_values) */
pqo_result pqopt_rle_dict_lookup_int64(int64_t* dst, uint64_t num_values, const uint8_t* src, uint64_t n, const int64_t* dict,
                                       uint64_t dict_len);

/* mixed-run level stream in parquet-mr style (arbitrary run lengths) */
uint64_t pqo_rle_encode_levels_mr(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, uint32_t bit_width);

#ifdef __cplusplus
}
#endif
#endif
