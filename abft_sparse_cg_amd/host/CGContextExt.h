// CGContextExt.h -- what this repository's driver (host/cg.cpp) may ask of a backend beyond
// the reference's thirteen virtuals.  A separate interface, reached by dynamic_cast, so that
// CGContext itself stays layout-compatible with the reference's (CGContext.h:8-67) and
// HIPContext.cpp still compiles against the reference header; the reference driver never
// looks for it.
#pragma once
#include <cstdint>

struct cg_matrix;
struct cg_vector;

class CGContextExt
{
public:
  virtual ~CGContextExt() {}

  // this process's place in a row-partitioned job (one process: 0 of 1)
  virtual int ext_rank() = 0;
  virtual int ext_size() = 0;

  // create_matrix for a driver that generated only this rank's row block: triplets of rows
  // [row_bounds[rank], row_bounds[rank+1]) (global indices, sorted), which are elements
  // [elem0, elem0 + count) of the whole matrix of nnz_total elements; row_bounds has
  // ext_size() + 1 entries.  NULL: not supported (format / single process).
  virtual cg_matrix* create_matrix_rows(const uint32_t *columns, const uint32_t *rows, const double *values, int N,
                                        long long nnz_total, const long long *row_bounds, long long elem0,
                                        long long count) = 0;

  // The loop of cg.cpp:87-118 for a fixed iteration count (-c 0) with alpha and beta kept on
  // the device, `blocks` times over: r = b, p = r, `warmup` untimed iterations, then `steps` timed
  // ones (bracketed by a device synchronisation and a barrier across ranks on both sides;
  // seconds[k] = the slowest rank's time for block k, `blocks` entries).  *rr = r.r after the
  // warmup + steps iterations of the last block (every block restarts the solve, so the number
  // of blocks does not change it).  ECC events are reported once, at the end.  false: not supported.
  virtual bool run_fixed(cg_matrix *A, cg_vector *b, cg_vector *x, cg_vector *r, cg_vector *p, cg_vector *w,
                         int warmup, int steps, int blocks, double *seconds, double *rr) = 0;
};
