/*
 * abft_oracle.c -- CPU restatement of the abft-sparse-cg hot path (see
 * abft_oracle.h: TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * Written from the behavioural description of the reference, not from its
 * text: the Hamming masks are generated from the construction rule, the six
 * spmv variants share one element-check routine, and events are queued
 * instead of printed.  Differences from the reference that are deliberate:
 *   - no undefined behaviour: elements are handled as word arrays, so a
 *     corrected VALUE bit is really written back (the reference needs
 *     -fno-strict-aliasing for that);
 *   - a gather index outside the input vector reads 0.0 instead of faulting;
 *   - a "corrected bit" >= 96 on a CSR element (possible only after a
 *     mis-decoded multi-bit error in sec7) is reported but not applied -- the
 *     reference writes past its 12-byte struct there;
 *   - trailing empty rows get rowptr = nnz (reference leaves them
 *     uninitialised, CSR/CPUContext.cpp:36-41).
 * Build: oracle/Makefile (gcc -O3 -ffp-contract=off -fopenmp).
 */
#include "abft_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define COLMASK 0x00FFFFFFu

struct ora_matrix {
  int fmt, mode, n_out, n_in, nnz;
  uint32_t index_base;
  uint32_t *cols, *rowptr; /* CSR */
  double *values;          /* CSR */
  uint32_t *elems;         /* COO: 4 words per element {col,row,val_lo,val_hi} */
  ora_event *ev;
  int nev, capev, fatal;
};

/* ------------------------------------------------------------------ ECC -- */

static int words_of(int fmt) { return fmt == ORA_FMT_CSR ? 3 : 4; }
static int eccword_of(int fmt) { return fmt == ORA_FMT_CSR ? 2 : 0; }
static int pow2(uint32_t x) { return x && !(x & (x - 1)); }

/* Construction rule stated in COO/ecc.h:136-170 (and, with the ECC byte in
 * word 2, reproducing the 21 constants of CSR/ecc.h:13-39): walk the bits in
 * word order handing out Hamming positions 3,5,6,7,9,... (powers of two are
 * skipped) to every bit except bits 24-31 of the ECC-carrying word; data bit
 * with position x is in mask p iff bit p-1 of x is set; check bit p lives at
 * bit 32-p of the ECC word and is in mask p only. */
void ora_ecc_masks(int fmt, uint32_t out[7][4]) {
  int nw = words_of(fmt), ew = eccword_of(fmt);
  for (int p = 1; p <= 7; p++) {
    uint32_t x = 3;
    for (int w = 0; w < 4; w++) out[p - 1][w] = 0;
    for (int w = 0; w < nw; w++) {
      for (int b = 0; b < 32; b++) {
        if (pow2(x)) x++;
        if (w == ew && b >= 24) {
          if (32 - b == p) out[p - 1][w] |= 1u << b;
        } else {
          if (x & (1u << (p - 1))) out[p - 1][w] |= 1u << b;
          x++;
        }
      }
    }
  }
}

static uint32_t g_mask[2][7][4];
static int g_mask_ready;
static void masks_init(void) {
  if (g_mask_ready) return;
  ora_ecc_masks(ORA_FMT_CSR, g_mask[ORA_FMT_CSR]);
  ora_ecc_masks(ORA_FMT_COO, g_mask[ORA_FMT_COO]);
  g_mask_ready = 1;
}

/* CSR/ecc.h:51-81, COO/ecc.h:63-101: check bit p (1..7) -> result bit 32-p */
uint32_t ora_ecc_syndrome(int fmt, const uint32_t *w) {
  masks_init();
  uint32_t s = 0;
  int nw = words_of(fmt);
  for (int p = 1; p <= 7; p++) {
    uint32_t acc = 0;
    for (int k = 0; k < nw; k++) acc ^= w[k] & g_mask[fmt][p - 1][k];
    s |= (uint32_t)__builtin_parity(acc) << (32 - p);
  }
  return s;
}

/* CSR/ecc.h:89-93, COO/ecc.h:109-113 */
uint32_t ora_ecc_parity(int fmt, const uint32_t *w) {
  uint32_t acc = 0;
  for (int k = 0; k < words_of(fmt); k++) acc ^= w[k];
  return (uint32_t)__builtin_parity(acc);
}

/* CSR/ecc.h:97-113, COO/ecc.h:117-134: syndrome -> index of the flipped bit.
 * The Hamming position h is rebuilt from the 7 syndrome bits; a power of two
 * is a check bit (bit 32-p of the ECC word), anything else is the
 * (h - floor(log2 h) - 2)-th data bit, and in COO data bits above 23 sit
 * behind the ECC byte. */
uint32_t ora_ecc_flipped_bit(int fmt, uint32_t syndrome) {
  uint32_t h = 0;
  for (int p = 1; p <= 7; p++)
    if ((syndrome >> (32 - p)) & 1u) h |= 1u << (p - 1);
  if (h == 0) return 0xFFFFFFFFu; /* reference: clz(0), undefined */
  int lg = 31 - __builtin_clz(h);
  if (pow2(h)) return (uint32_t)(31 - lg) + (fmt == ORA_FMT_CSR ? 64u : 0u);
  uint32_t d = h - (uint32_t)lg - 2u;
  if (fmt == ORA_FMT_COO && d >= 24u) d += 8u;
  return d;
}

/* generate_ecc_bits: CSR/CPUContext.cpp:7-9,209-212,247-250,291-295,347-351;
 * COO/CPUContext.cpp:7-9,196-199,234-237,277-281,330-334 */
void ora_ecc_encode(int fmt, int mode, uint32_t *w) {
  int ew = eccword_of(fmt);
  switch (mode) {
    case ORA_MODE_SED: w[ew] |= ora_ecc_parity(fmt, w) << 31; break;
    case ORA_MODE_SEC7: w[ew] |= ora_ecc_syndrome(fmt, w); break;
    case ORA_MODE_SEC8:
    case ORA_MODE_SECDED:
      w[ew] |= ora_ecc_syndrome(fmt, w);
      w[ew] |= ora_ecc_parity(fmt, w) << 24;
      break;
    default: break;
  }
}

uint32_t ora_csr_encode_col(int mode, uint64_t vb, uint32_t col) {
  uint32_t w[3] = {(uint32_t)vb, (uint32_t)(vb >> 32), col};
  ora_ecc_encode(ORA_FMT_CSR, mode, w);
  return w[2];
}
uint32_t ora_coo_encode_col(int mode, uint32_t col, uint32_t row, uint64_t vb) {
  uint32_t w[4] = {col, row, (uint32_t)vb, (uint32_t)(vb >> 32)};
  ora_ecc_encode(ORA_FMT_COO, mode, w);
  return w[0];
}

/* --------------------------------------------------------------- events -- */

int ora_event_is_fatal(uint32_t k) {
  return k == ORA_EV_SED_DETECTED || k == ORA_EV_DOUBLE_BIT || k >= ORA_EV_ROW_SIZE;
}

static void push_event(ora_matrix *m, uint32_t kind, uint32_t index, uint32_t bit) {
#pragma omp critical(ora_events)
  {
    if (m->nev == m->capev) {
      m->capev = m->capev ? 2 * m->capev : 64;
      m->ev = (ora_event *)realloc(m->ev, (size_t)m->capev * sizeof(ora_event));
    }
    ora_event e = {kind, index, bit, (uint32_t)m->fmt};
    m->ev[m->nev++] = e;
    if (ora_event_is_fatal(kind)) m->fatal = 1;
  }
}

static int ev_cmp(const void *a, const void *b) {
  const ora_event *x = (const ora_event *)a, *y = (const ora_event *)b;
  if (x->index != y->index) return x->index < y->index ? -1 : 1;
  if (x->kind != y->kind) return x->kind < y->kind ? -1 : 1;
  return 0;
}

int ora_events(ora_matrix *m, ora_event *buf, int cap, int *fatal) {
  if (m->nev > 0) qsort(m->ev, (size_t)m->nev, sizeof(ora_event), ev_cmp);  /* (no events: the array may not exist yet) */
  int n = 0, f = 0;
  for (int i = 0; i < m->nev && n < cap; i++) {
    buf[n++] = m->ev[i];
    if (ora_event_is_fatal(m->ev[i].kind)) { f = 1; break; }
  }
  m->nev = 0;
  m->fatal = 0;
  if (fatal) *fatal = f;
  return n;
}

int ora_format_event(const ora_event *e, char *buf, size_t cap) {
  int coo = e->fmt == ORA_FMT_COO;
  switch (e->kind) {
    case ORA_EV_SED_DETECTED: /* CSR/CPUContext.cpp:233, COO:215 */
      return snprintf(buf, cap, "[ECC] error detected at index %d\n", (int)e->index);
    case ORA_EV_CORRECTED_BIT: /* CSR :278,:324,:380; COO :257,:300,:354 */
      return snprintf(buf, cap, "[ECC] corrected bit %u at index %d\n", e->bit, (int)e->index);
    case ORA_EV_CORRECTED_PARITY: /* CSR :331,:387; COO :307,:361 */
      return snprintf(buf, cap, "[ECC] corrected overall parity bit at index %d\n", (int)e->index);
    case ORA_EV_DOUBLE_BIT: /* CSR :398, COO :371 */
      return snprintf(buf, cap, "[ECC] double-bit error detected\n");
    case ORA_EV_ROW_SIZE: /* CSR :175, COO :158 */
      return snprintf(buf, cap, coo ? "row size constraint violated for index %d\n"
                                    : "row size constraint violated for row %d\n", (int)e->index);
    case ORA_EV_ROW_ORDER: /* CSR :180 (sic, no space), COO :175 */
      return snprintf(buf, cap, coo ? "row index order violated at index %d\n"
                                    : "row order constraint violated for row%d\n", (int)e->index);
    case ORA_EV_COL_SIZE: /* CSR :190, COO :163 */
      return snprintf(buf, cap, coo ? "column size constraint violated for index %d\n"
                                    : "column size constraint violated at index %d\n", (int)e->index);
    case ORA_EV_COL_ORDER: /* CSR :197, COO :184 */
      return snprintf(buf, cap, coo ? "column index order violated at index %d\n"
                                    : "column order constraint violated at index %d\n", (int)e->index);
    default: return snprintf(buf, cap, "unknown event %u\n", e->kind);
  }
}

/* --------------------------------------------------------------- matrix -- */

ora_matrix *ora_matrix_create(int fmt, int mode, const uint32_t *cols, const uint32_t *rows,
                              const double *vals, int n_out, int n_in, int nnz,
                              uint32_t index_base) {
  ora_matrix *m = (ora_matrix *)calloc(1, sizeof(*m));
  m->fmt = fmt; m->mode = mode; m->n_out = n_out; m->n_in = n_in; m->nnz = nnz;
  m->index_base = index_base;
  if (fmt == ORA_FMT_CSR) {
    /* CSR/CPUContext.cpp:11-44 */
    m->cols = (uint32_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * 4);
    m->values = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * 8);
    m->rowptr = (uint32_t *)malloc((size_t)(n_out + 1) * 4);
    uint32_t next = 0;
    for (int i = 0; i < nnz; i++) {
      uint32_t w[3];
      memcpy(w, &vals[i], 8);
      w[2] = cols[i];
      ora_ecc_encode(fmt, mode, w);
      memcpy(&m->values[i], w, 8);
      m->cols[i] = w[2];
      while (next <= rows[i] && next <= (uint32_t)n_out) m->rowptr[next++] = (uint32_t)i;
    }
    while (next <= (uint32_t)n_out) m->rowptr[next++] = (uint32_t)nnz;
    m->rowptr[n_out] = (uint32_t)nnz;
  } else {
    /* COO/CPUContext.cpp:11-36 */
    m->elems = (uint32_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * 16);
    for (int i = 0; i < nnz; i++) {
      uint32_t *w = m->elems + 4 * (size_t)i;
      w[0] = cols[i];
      w[1] = rows[i];
      memcpy(w + 2, &vals[i], 8);
      ora_ecc_encode(fmt, mode, w);
    }
  }
  return m;
}

void ora_matrix_destroy(ora_matrix *m) {
  if (!m) return;
  free(m->cols); free(m->rowptr); free(m->values); free(m->elems); free(m->ev); free(m);
}
uint32_t *ora_matrix_csr_cols(ora_matrix *m) { return m->cols; }
uint32_t *ora_matrix_csr_rowptr(ora_matrix *m) { return m->rowptr; }
double *ora_matrix_csr_values(ora_matrix *m) { return m->values; }
void *ora_matrix_coo_elements(ora_matrix *m) { return m->elems; }

/* CSR/CPUContext.cpp:146-158, COO/CPUContext.cpp:134-139 */
void ora_inject(ora_matrix *m, uint32_t index, const int *bits, int nbits) {
  for (int k = 0; k < nbits; k++) {
    int bit = bits[k];
    uint32_t x = 1u << (bit % 32);
    if (m->fmt == ORA_FMT_CSR) {
      if (bit < 64) {
        uint32_t w[2];
        memcpy(w, &m->values[index], 8);
        w[bit / 32] ^= x;
        memcpy(&m->values[index], w, 8);
      } else {
        m->cols[index] ^= x;
      }
    } else {
      m->elems[4 * (size_t)index + bit / 32] ^= x;
    }
  }
}

/* CSR/CPUContext.cpp:137-148, COO/CPUContext.cpp:125-136: 1 + num_flips
 * rand() calls, in this order */
int ora_inject_rand(ora_matrix *m, int kind, int num_flips, int *bits_out) {
  int index = rand() % m->nnz;
  int start = 0, end = m->fmt == ORA_FMT_CSR ? 96 : 128;
  if (m->fmt == ORA_FMT_CSR) {
    if (kind == 1) end = 64; else if (kind == 2) start = 64;
  } else {
    if (kind == 1) start = 64; else if (kind == 2) end = 64;
  }
  for (int i = 0; i < num_flips; i++) {
    int bit = rand() % (end - start) + start;
    if (bits_out) bits_out[i] = bit;
    ora_inject(m, (uint32_t)index, &bit, 1);
  }
  return index;
}

/* ----------------------------------------------------------------- spmv -- */

/* The per-element ECC step shared by all modes and both formats: checks the
 * codeword `w`, repairs it in place when the mode can, queues the event.
 * Returns 1 if the element was modified (caller writes it back), -1 on a
 * fatal event, else 0.
 *   sed    CSR/CPUContext.cpp:230-235   COO :212-217
 *   sec7   CSR :268-279                 COO :250-258
 *   sec8   CSR :313-335 (lazy syndrome) COO :293-312
 *   secded CSR :369-400                 COO :346-373 */
static int check_element(ora_matrix *m, uint32_t *w, uint32_t gidx) {
  int fmt = m->fmt, nbits = 32 * words_of(fmt), ew = eccword_of(fmt);
  switch (m->mode) {
    case ORA_MODE_SED:
      if (ora_ecc_parity(fmt, w)) { push_event(m, ORA_EV_SED_DETECTED, gidx, 0); return -1; }
      return 0;
    case ORA_MODE_SEC7: {
      uint32_t s = ora_ecc_syndrome(fmt, w);
      if (!s) return 0;
      uint32_t bit = ora_ecc_flipped_bit(fmt, s);
      if (bit < (uint32_t)nbits) w[bit / 32] ^= 1u << (bit % 32);
      push_event(m, ORA_EV_CORRECTED_BIT, gidx, bit);
      return 1;
    }
    case ORA_MODE_SEC8:
    case ORA_MODE_SECDED: {
      uint32_t par = ora_ecc_parity(fmt, w);
      if (m->mode == ORA_MODE_SEC8 && !par) return 0;
      uint32_t s = ora_ecc_syndrome(fmt, w);
      if (par) {
        if (s) {
          uint32_t bit = ora_ecc_flipped_bit(fmt, s);
          if (bit < (uint32_t)nbits) w[bit / 32] ^= 1u << (bit % 32);
          push_event(m, ORA_EV_CORRECTED_BIT, gidx, bit);
        } else {
          w[ew] ^= 1u << 24;
          push_event(m, ORA_EV_CORRECTED_PARITY, gidx, 0);
        }
        return 1;
      }
      if (s) { push_event(m, ORA_EV_DOUBLE_BIT, gidx, 0); return -1; }
      return 0;
    }
    default: return 0;
  }
}

static inline double gather(const double *x, uint32_t idx, int n) {
  return idx < (uint32_t)n ? x[idx] : 0.0;
}

/* CSR/CPUContext.cpp:115-133 and :162-207,:214-245,:252-289,:297-345,:353-411 */
static void spmv_csr(ora_matrix *m, const double *x, double *y, int threads) {
  const int ecc = m->mode >= ORA_MODE_SED;
  volatile int *fatal = &m->fatal;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int row = 0; row < m->n_out; row++) {
    if (*fatal) continue;
    uint32_t start = m->rowptr[row], end = m->rowptr[row + 1];
    if (m->mode == ORA_MODE_CONSTRAINTS) {
      if (end > (uint32_t)m->nnz) { push_event(m, ORA_EV_ROW_SIZE, (uint32_t)row, 0); continue; }
      if (end < start) { push_event(m, ORA_EV_ROW_ORDER, (uint32_t)row, 0); continue; }
    }
    double tmp = 0.0;
    int dead = 0;
    for (uint32_t i = start; i < end && !dead; i++) {
      uint32_t col = m->cols[i];
      if (m->mode == ORA_MODE_CONSTRAINTS) {
        if (col >= (uint32_t)m->n_in) {
          push_event(m, ORA_EV_COL_SIZE, m->index_base + i, 0); dead = 1; break;
        }
        if (i < end - 1 && m->cols[i + 1] <= col) {
          push_event(m, ORA_EV_COL_ORDER, m->index_base + i, 0); dead = 1; break;
        }
      } else if (ecc) {
        uint32_t w[3];
        memcpy(w, &m->values[i], 8);
        w[2] = col;
        int rc = check_element(m, w, m->index_base + i);
        if (rc < 0) { dead = 1; break; }
        if (rc > 0) { memcpy(&m->values[i], w, 8); m->cols[i] = w[2]; }
        col = w[2] & COLMASK;
      }
      tmp += m->values[i] * gather(x, col, m->n_in);
    }
    if (!dead) y[row] = tmp;
  }
}

/* COO/CPUContext.cpp:104-121 and :142-190,:201-227,:239-270,:283-323,:336-379:
 * serial scatter in storage order */
static void spmv_coo(ora_matrix *m, const double *x, double *y) {
  const int ecc = m->mode >= ORA_MODE_SED;
  for (int i = 0; i < m->n_out; i++) y[i] = 0.0;
  for (int i = 0; i < m->nnz; i++) {
    uint32_t *w = m->elems + 4 * (size_t)i;
    uint32_t col = w[0], row = w[1];
    uint32_t gi = m->index_base + (uint32_t)i;
    if (m->mode == ORA_MODE_CONSTRAINTS) {
      if (row >= (uint32_t)m->n_in) { push_event(m, ORA_EV_ROW_SIZE, gi, 0); return; }
      if (col >= (uint32_t)m->n_out) { push_event(m, ORA_EV_COL_SIZE, gi, 0); return; }
      if (i < m->nnz - 1) {
        uint32_t ncol = w[4], nrow = w[5];
        if (row > nrow) { push_event(m, ORA_EV_ROW_ORDER, gi, 0); return; }
        if (row == nrow && col >= ncol) { push_event(m, ORA_EV_COL_ORDER, gi, 0); return; }
      }
    } else if (ecc) {
      uint32_t e[4] = {w[0], w[1], w[2], w[3]};
      int rc = check_element(m, e, gi);
      if (rc < 0) return;
      if (rc > 0) memcpy(w, e, 16);
      col = e[0] & COLMASK;
      row = e[1];
    }
    double v;
    memcpy(&v, w + 2, 8);
    if (col < (uint32_t)m->n_out) y[col] += v * gather(x, row, m->n_in);
  }
}

int ora_spmv(ora_matrix *m, const double *x, double *y, int threads) {
  int before = m->nev;
  if (threads < 1) threads = 1;
  if (m->fmt == ORA_FMT_CSR) spmv_csr(m, x, y, threads); else spmv_coo(m, x, y);
  return m->nev - before;
}

/* --------------------------------------------------------- vector kernels -- */

/* CSR/CPUContext.cpp:82-90: strict left-to-right sum, separate mul and add */
double ora_dot(const double *a, const double *b, int n) {
  double ret = 0.0;
  for (int i = 0; i < n; i++) ret += a[i] * b[i];
  return ret;
}

/* CSR/CPUContext.cpp:92-105 */
double ora_calc_xr(double *x, double *r, const double *p, const double *w, double alpha, int n) {
  double ret = 0.0;
  for (int i = 0; i < n; i++) {
    x[i] += alpha * p[i];
    r[i] -= alpha * w[i];
    ret += r[i] * r[i];
  }
  return ret;
}

/* CSR/CPUContext.cpp:107-113 */
void ora_calc_p(double *p, const double *r, double beta, int n) {
  for (int i = 0; i < n; i++) p[i] = r[i] + beta * p[i];
}

/* cg.cpp:87-118 (square system: n_out == n_in) */
int ora_cg(ora_matrix *m, const double *b, double *x, double *r, double *p, double *w,
           int max_itrs, double conv, double *rr_hist, int threads, int *fatal) {
  int n = m->n_out;
  memcpy(r, b, (size_t)n * 8);
  memcpy(p, r, (size_t)n * 8);
  double rr = ora_dot(r, r, n);
  int itr = 0;
  if (fatal) *fatal = 0;
  for (; itr < max_itrs && rr > conv; itr++) {
    ora_spmv(m, p, w, threads);
    if (m->fatal) { if (fatal) *fatal = 1; break; }
    double pw = ora_dot(p, w, n);
    double alpha = rr / pw;
    double rr_new = ora_calc_xr(x, r, p, w, alpha, n);
    double beta = rr_new / rr;
    ora_calc_p(p, r, beta, n);
    rr = rr_new;
    if (rr_hist) rr_hist[itr] = rr;
  }
  return itr;
}
