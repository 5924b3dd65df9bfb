// ref_harness.cpp -- thin C ABI over the REFERENCE's own CPUContext objects.
//
// TEST INFRASTRUCTURE.  This file contains no reference code: it is compiled
// by oracle/Makefile together with the reference sources where they lie
// (/root/reference/CGContext.cpp and {CSR,COO}/CPUContext.cpp) into
// oracle/_ref/libref_{csr,coo}.so, so that tests can drive the real reference
// backend in-process and compare it with oracle/abft_oracle.c, and so that
// bench.py can time the real reference as the CPU baseline.
//
// The reference prints its ECC events with printf and calls exit(1) on fatal
// ones; callers that need those run the call in a forked child with fd 1
// redirected (tests/_capture.py).
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "CPUContext.h"  // -I /root/reference/{CSR|COO}: cg_matrix, cg_vector, (csr|coo)_element

extern "C" {

void *ref_create(const char *mode) { return CGContext::create("cpu", mode); }
void ref_destroy(void *ctx) { delete (CGContext *)ctx; }

void *ref_matrix_create(void *ctx, const uint32_t *cols, const uint32_t *rows,
                        const double *vals, int N, int nnz) {
  return ((CGContext *)ctx)->create_matrix(cols, rows, vals, N, nnz);
}
void ref_matrix_destroy(void *ctx, void *mat) {
  ((CGContext *)ctx)->destroy_matrix((cg_matrix *)mat);
}

#if REF_FMT == 0
// CSR: copy out the stored arrays (any pointer may be NULL)
void ref_matrix_read(void *mat, uint32_t *cols, uint32_t *rowptr, double *values) {
  cg_matrix *m = (cg_matrix *)mat;
  if (cols) memcpy(cols, m->cols, (size_t)m->nnz * 4);
  if (rowptr) memcpy(rowptr, m->rows, ((size_t)m->N + 1) * 4);
  if (values) memcpy(values, m->values, (size_t)m->nnz * 8);
}
// XOR one bit of element `index`, numbered as the reference's inject_bitflip
// numbers them (0-63 value, 64-95 column)
void ref_flip(void *mat, uint32_t index, int bit) {
  cg_matrix *m = (cg_matrix *)mat;
  if (bit < 64) {
    uint32_t w[2];
    memcpy(w, &m->values[index], 8);
    w[bit / 32] ^= 1u << (bit % 32);
    memcpy(&m->values[index], w, 8);
  } else {
    m->cols[index] ^= 1u << (bit % 32);
  }
}
#else
// COO: 16-byte elements
void ref_matrix_read(void *mat, void *elements, uint32_t *, double *) {
  cg_matrix *m = (cg_matrix *)mat;
  if (elements) memcpy(elements, m->elements, (size_t)m->nnz * 16);
}
void ref_flip(void *mat, uint32_t index, int bit) {
  cg_matrix *m = (cg_matrix *)mat;
  uint32_t w[4];
  memcpy(w, &m->elements[index], 16);
  w[bit / 32] ^= 1u << (bit % 32);
  memcpy(&m->elements[index], w, 16);
}
#endif

void ref_inject_rand(void *ctx, void *mat, int kind, int num_flips) {
  ((CGContext *)ctx)->inject_bitflip((cg_matrix *)mat, (CGContext::BitFlipKind)kind, num_flips);
}

void ref_spmv(void *ctx, void *mat, const double *x, double *y, int N) {
  cg_vector vx = {N, const_cast<double *>(x)};
  cg_vector vy = {N, y};
  ((CGContext *)ctx)->spmv((cg_matrix *)mat, &vx, &vy);
}
double ref_dot(void *ctx, const double *a, const double *b, int N) {
  cg_vector va = {N, const_cast<double *>(a)}, vb = {N, const_cast<double *>(b)};
  return ((CGContext *)ctx)->dot(&va, &vb);
}
double ref_calc_xr(void *ctx, double *x, double *r, const double *p, const double *w,
                   double alpha, int N) {
  cg_vector vx = {N, x}, vr = {N, r}, vp = {N, const_cast<double *>(p)},
            vw = {N, const_cast<double *>(w)};
  return ((CGContext *)ctx)->calc_xr(&vx, &vr, &vp, &vw, alpha);
}
void ref_calc_p(void *ctx, double *p, const double *r, double beta, int N) {
  cg_vector vp = {N, p}, vr = {N, const_cast<double *>(r)};
  ((CGContext *)ctx)->calc_p(&vp, &vr, beta);
}

// The CG loop, calling the reference backend in the order the reference
// driver does (cg.cpp:87-118); the per-iteration printf is left out.
int ref_cg(void *ctxv, void *mat, const double *b, double *x, double *r, double *p, double *w,
           int N, int max_itrs, double conv, double *rr_hist) {
  CGContext *ctx = (CGContext *)ctxv;
  cg_vector vb = {N, const_cast<double *>(b)}, vx = {N, x}, vr = {N, r}, vp = {N, p},
            vw = {N, w};
  ctx->copy_vector(&vr, &vb);
  ctx->copy_vector(&vp, &vr);
  double rr = ctx->dot(&vr, &vr);
  int itr = 0;
  for (; itr < max_itrs && rr > conv; itr++) {
    ctx->spmv((cg_matrix *)mat, &vp, &vw);
    double pw = ctx->dot(&vp, &vw);
    double alpha = rr / pw;
    double rr_new = ctx->calc_xr(&vx, &vr, &vp, &vw, alpha);
    double beta = rr_new / rr;
    ctx->calc_p(&vp, &vr, beta);
    rr = rr_new;
    if (rr_hist) rr_hist[itr] = rr;
  }
  return itr;
}

// The reference's bit-level ECC functions (static inline in its ecc.h), on a
// raw word image of one element: 3 words CSR {val_lo,val_hi,col}, 4 words COO
// {col,row,val_lo,val_hi}.
#if REF_FMT == 0
typedef csr_element ref_element;
#define REF_WORDS 3
#else
typedef coo_element ref_element;
#define REF_WORDS 4
#endif
uint32_t ref_ecc_syndrome(const uint32_t *w) {
  ref_element e;
  memcpy(&e, w, 4 * REF_WORDS);
  return ecc_compute_col8(e);
}
uint32_t ref_ecc_parity(const uint32_t *w) {
  ref_element e;
  memcpy(&e, w, 4 * REF_WORDS);
  return ecc_compute_overall_parity(e);
}
uint32_t ref_ecc_flipped_bit(uint32_t syndrome) { return ecc_get_flipped_bit_col8(syndrome); }

void ref_flush(void) { fflush(stdout); }

}  // extern "C"
