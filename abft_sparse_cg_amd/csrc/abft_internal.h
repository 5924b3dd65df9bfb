// abft_internal.h -- shared between kernels.hip (device code + launchers) and
// abft_hip.hip (the C ABI).  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/abft_hip.h"

// Device-side event queue: one entry per printf line of the reference backend.
struct EventRing {
  abft_event *buf;
  uint32_t *count;  // total pushed (may exceed cap; extra events are dropped)
  uint32_t cap;
};

// CSR matrix as the kernels see it.  cols/vals keep the reference's SoA layout
// (CSR/CPUContext.h:11-18) so element i is {vals[i], cols[i]}; both arrays are
// over-allocated by 2 elements so the paired loads never leave the buffer.
struct CsrDev {
  uint32_t *cols;
  double *vals;
  const uint32_t *rowptr;   // n_out + 1
  const uint32_t *blk_row;  // nblk + 1 : first row of each row block
  uint32_t nblk, n_out, n_in, nnz, index_base;
};

// COO matrix: 16-byte elements {col,row,value} (COO/ecc.h:11-16) stored grouped
// by output index (col) and, inside a group, in the caller's order -- so each
// output is still summed in the reference's storage order.
struct CooDev {
  uint4 *elems;
  const uint32_t *grp_ptr;      // n_out + 1
  const uint32_t *blk_grp;      // nblk + 1
  const uint32_t *orig_index;   // stored position -> caller's element index (cold path)
  const uint32_t *pos_of_orig;  // caller's element index -> stored position
  uint32_t nblk, n_out, n_in, nnz, index_base;
};

#define ABFT_COLMASK_HOST 0x00FFFFFFu

// Tuning knobs (overridable with -D for A/B builds; defaults are the measured best)
#ifndef ABFT_CFG_CSR_EPT
#define ABFT_CFG_CSR_EPT 4
#endif
#ifndef ABFT_CFG_COO_EPT
#define ABFT_CFG_COO_EPT 4
#endif
#ifndef ABFT_CFG_NT
#define ABFT_CFG_NT 1  // stream cols/vals with the non-temporal hint
#endif
#ifndef ABFT_CFG_XCD
#define ABFT_CFG_XCD 1  // XCD-aware tile order
#endif

constexpr int ABFT_BLOCK = 256;
constexpr int ABFT_CSR_EPT = ABFT_CFG_CSR_EPT;           // elements per thread per tile
constexpr int ABFT_CSR_TILE = ABFT_BLOCK * ABFT_CSR_EPT;  // nnz staged per block
constexpr int ABFT_COO_EPT = ABFT_CFG_COO_EPT;
constexpr int ABFT_COO_TILE = ABFT_BLOCK * ABFT_COO_EPT;
constexpr int ABFT_MAX_PARTIALS = 2048;  // reduction blocks (256 CUs x 8)

hipError_t launch_encode_csr(int mode, uint32_t *cols, double *vals, uint32_t nnz, hipStream_t s);
hipError_t launch_encode_coo(int mode, uint4 *elems, uint32_t nnz, hipStream_t s);
hipError_t launch_spmv_csr(int mode, const CsrDev &A, const double *x, double *y, EventRing ev,
                           hipStream_t s);
hipError_t launch_spmv_coo(int mode, const CooDev &A, const double *x, double *y, EventRing ev,
                           hipStream_t s);
hipError_t launch_inject_csr(double *vals, uint32_t *cols, uint32_t index, const int *bits_dev,
                             int nbits, hipStream_t s);
hipError_t launch_inject_coo(uint4 *elems, const uint32_t *pos_of_orig, uint32_t index,
                             const int *bits_dev, int nbits, hipStream_t s);

// Reductions: stage 1 writes one partial per block into `partials`
// (ABFT_MAX_PARTIALS doubles), stage 2 sums them in a fixed order and writes
// the scalar to dev_out and/or host_out (pinned, device-visible), together
// with the event count when ev_count/host_evcount are given.
int reduce_blocks(int n);
hipError_t launch_dot(const double *a, const double *b, int n, double *partials, hipStream_t s);
hipError_t launch_calc_xr(double *x, double *r, const double *p, const double *w, double alpha,
                          int n, double *partials, hipStream_t s);
hipError_t launch_finalize(const double *partials, int nparts, double *dev_out, double *host_out,
                           const uint32_t *ev_count, uint32_t *host_evcount, hipStream_t s);
hipError_t launch_calc_p(double *p, const double *r, double beta, int n, hipStream_t s);
hipError_t launch_stream_copy(double *dst, const double *src, size_t n, hipStream_t s);
hipError_t launch_stream_read(const double *src, size_t n, double *sink, hipStream_t s);
