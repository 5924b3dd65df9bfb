/*
 * abft_oracle.h -- CPU restatement of the abft-sparse-cg hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped HIP path never
 * does.  Every function cites the reference file:line whose behaviour it
 * restates (reference tree = DataIntensive-HPC/abft-sparse-cg).
 *
 * Parity pinning: tests/test_oracle_vs_ref.py compares this library with the
 * reference's own CPUContext objects compiled from /root/reference by
 * oracle/Makefile into oracle/_ref/ (with -fno-strict-aliasing),
 * and tests/golden/ holds vectors generated from that build.
 */
#ifndef ABFT_ORACLE_H
#define ABFT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same numbering as include/abft_hip.h */
enum { ORA_MODE_NONE = 0, ORA_MODE_CONSTRAINTS, ORA_MODE_SED, ORA_MODE_SEC7, ORA_MODE_SEC8,
       ORA_MODE_SECDED };
enum { ORA_FMT_CSR = 0, ORA_FMT_COO = 1 };
enum { ORA_EV_SED_DETECTED = 1, ORA_EV_CORRECTED_BIT, ORA_EV_CORRECTED_PARITY,
       ORA_EV_DOUBLE_BIT, ORA_EV_ROW_SIZE, ORA_EV_ROW_ORDER, ORA_EV_COL_SIZE,
       ORA_EV_COL_ORDER };

typedef struct { uint32_t kind, index, bit, fmt; } ora_event;
typedef struct ora_matrix ora_matrix;

/* ---- bit-level ECC (CSR/ecc.h, COO/ecc.h) ---- */
void     ora_ecc_masks(int fmt, uint32_t out[7][4]);                /* COO/ecc.h:136-170 rule */
uint32_t ora_ecc_syndrome(int fmt, const uint32_t *words);          /* CSR/ecc.h:51-81, COO/ecc.h:63-101 */
uint32_t ora_ecc_parity(int fmt, const uint32_t *words);            /* CSR/ecc.h:89-93, COO/ecc.h:109-113 */
uint32_t ora_ecc_flipped_bit(int fmt, uint32_t syndrome);           /* CSR/ecc.h:97-113, COO/ecc.h:117-134 */
void     ora_ecc_encode(int fmt, int mode, uint32_t *words);        /* generate_ecc_bits, all modes */
uint32_t ora_csr_encode_col(int mode, uint64_t value_bits, uint32_t col);
uint32_t ora_coo_encode_col(int mode, uint32_t col, uint32_t row, uint64_t value_bits);

/* ---- matrix (CSR/CPUContext.cpp:11-52, COO/CPUContext.cpp:11-42) ---- */
ora_matrix *ora_matrix_create(int fmt, int mode, const uint32_t *cols, const uint32_t *rows,
                              const double *vals, int nrows, int ncols, int nnz,
                              uint32_t index_base);
void      ora_matrix_destroy(ora_matrix *m);
uint32_t *ora_matrix_csr_cols(ora_matrix *m);
uint32_t *ora_matrix_csr_rowptr(ora_matrix *m);
double   *ora_matrix_csr_values(ora_matrix *m);
void     *ora_matrix_coo_elements(ora_matrix *m); /* 16-byte {col,row,value} */

/* CSR/CPUContext.cpp:135-159, COO/CPUContext.cpp:123-140 */
void ora_inject(ora_matrix *m, uint32_t index, const int *bits, int nbits);
/* the same with the reference's libc rand() draws; returns the index drawn */
int  ora_inject_rand(ora_matrix *m, int kind, int num_flips, int *bits_out);

/* All six spmv variants of each format.  Returns the number of events queued
 * by this call; a fatal event stops the pass at that element, like exit(1). */
int ora_spmv(ora_matrix *m, const double *x, double *y, int threads);
int ora_events(ora_matrix *m, ora_event *buf, int cap, int *fatal); /* drain */

/* CSR/CPUContext.cpp:82-113 (identical in COO/CPUContext.cpp:71-102) */
double ora_dot(const double *a, const double *b, int n);
double ora_calc_xr(double *x, double *r, const double *p, const double *w, double alpha, int n);
void   ora_calc_p(double *p, const double *r, double beta, int n);

/* cg.cpp:87-118: returns iterations run; rr_hist (may be NULL) gets rr after
 * every iteration; stops early with *fatal=1 when an spmv raised a fatal event */
int ora_cg(ora_matrix *m, const double *b, double *x, double *r, double *p, double *w,
           int max_itrs, double conv_threshold, double *rr_hist, int threads, int *fatal);

/* the reference's exact printf line (with newline) for an event */
int ora_format_event(const ora_event *ev, char *buf, size_t cap);
int ora_event_is_fatal(uint32_t kind);

#ifdef __cplusplus
}
#endif
#endif
