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
  const uint4 *blk;         // nblk : {first row, end row, first element, end element} of each row block
  uint32_t nblk, n_out, n_in, nnz, index_base;
  // Panel layout only (see CsrPanels): elements are stored grouped by
  // (row group, column panel); these map storage position <-> caller's index.
  const uint32_t *orig_index;   // cold path: event messages carry the caller's element index
  const uint32_t *pos_of_orig;  // inject
  const uint32_t *gidx;         // shards whose elements are not one run of the caller's: global index of local element k
};

// Panel ("column-blocked") layout for matrices whose columns are scattered over
// a vector much larger than an XCD's 4 MB L2 (random / unstructured).  Rows are
// cut into groups of ABFT_PANEL_ROWS, columns into panels of `width` entries
// (2 MB of x); the elements of one (row group, panel) segment are contiguous and
// keep their (row, col) order.  A workgroup owns a row group, keeps the running
// row sums in registers and sweeps the panels in ascending order -- the same
// order of additions as the row-by-row sum, so y stays bit-identical -- while all
// resident workgroups gather from the same 2 MB window of x at a time, which the
// L2 then serves (the streaming layout misses L2 on 7 of 8 gathers here).
struct CsrPanels {
  const uint32_t *seg_base;  // ngroups * npanels + 1 : first element of each segment
  const uint16_t *seg_ptr;   // per segment ABFT_PANEL_ROWS + 1 row offsets relative to seg_base
  uint32_t ngroups, npanels;
  unsigned long long *debug;  // optional (ABFT_HIP_PANEL_DEBUG, -DABFT_DBG_STAMPS builds): 2 x 8 phase clocks, first / later launches
  uint32_t *pace;             // COO panel kernel run as ONE launch over all panels: a progress board per XCD (as SweepLayout), else NULL
  uint32_t lag;               // workgroups of an XCD stay within `lag` panels of its slowest; 0: no pacing
  uint32_t width;             // gather indices per panel
  uint32_t xpf;               // COO panel kernel: entering a panel, the workgroups of an XCD touch the NEXT panel's lines of x (into that XCD's L2)
};
// Sweep layout: the panel layout's successor (same idea: (output group, gather-index panel)
// segments, outputs' additions in the caller's order) run as ONE persistent launch:
//   * a workgroup owns a group of 256 * RPT outputs for the whole sweep; thread (wave w,
//     lane l) keeps the running sums of outputs w*64*RPT + j*64 + l (j < RPT) in
//     registers from the first panel to the last -- y is written once (the chunked panel
//     launches carried the sums through y: 7 extra round trips of y on config 4);
//   * per (segment, row) only a COUNT is stored (1 byte; a matrix with more than 255 elements
//     of one row in one panel keeps the panel layout), laid out so that a thread loads its
//     RPT counts with one access; the element offsets come from
//     a DPP prefix scan over the wave plus one stored base per (segment, wave)
//     (the 16-bit offset tables of the panel layout were 134 MB on config 4);
//   * what keeps every workgroup of an XCD inside the same window of the gathered vector is
//     not a kernel boundary but pacing through a per-XCD progress board (kernels.hip,
//     board_load): speed only, never correctness; every wait is bounded.
struct SweepLayout {
  const uint32_t *wbase;   // nseg * 4 + 1 : first element of (segment, wave); segment s = [wbase[4s], wbase[4s+4])
  const uint8_t *counts;   // [segment][thread 0..255][j 0..RPT-1], one byte each
  uint32_t ngroups, npanels;
  uint32_t *pace;          // exit ticket, registrations per XCD, a progress board per XCD (kernels.hip: PACE_*)
  uint32_t lag;            // workgroups of an XCD stay within `lag` panels of its slowest; 0: no pacing
  uint32_t *debug;         // optional (ABFT_HIP_SWEEP_DEBUG): {polls that waited, waits, workgroup exits}, never reset
  unsigned long long *debug_wg;  // optional, -DABFT_DBG_STAMPS builds: per workgroup {clocks in all, at the pacing check, staging, hardware id}
};

// Slice layout (round 3; opt-in, ABFT_HIP_LAYOUT=slice: built to test whether the sweep kernel's bookkeeping
// is what keeps it from the gather path's floor -- it is not, DESIGN.md section 4 "Round 3" -- and measured
// slower).  What the sweep kernel pays per (row, panel) CELL -- a count byte, a share of a prefix scan, a
// range test, a branch, and a workgroup barrier per tile -- is as much as it pays per element (config 4:
// 26 panels x 16 rows = 416 cells for 400 elements per thread).  Here nothing is per cell:
//   * rows are cut into SLICES of 2^rows_log2 consecutive rows; a slice belongs to ONE WAVE, which
//     keeps the slice's running row sums in its own piece of LDS -- no other wave ever touches them,
//     so the kernel has no barrier and no atomics at all;
//   * a slice's elements are stored as one run ordered by (panel of the gather index, row, caller's
//     order), each with its row inside the slice as a 16-bit id: the wave streams the run 64 elements
//     per instruction -- chunks run straight across panel boundaries, every chunk but a slice's last
//     is full -- and each element is added onto its row's sum in LDS, in storage order (the lanes
//     of one chunk that share a row are folded in lane order by the first of them, over DPP);
//     a row's elements therefore meet its sum in the caller's order (panels ascend; inside a panel a
//     row's elements keep their order): y stays bit-identical to the reference;
//   * `sub` (first element of every (slice, panel)) is only read for panel ranges and pacing.
struct SliceLayout {
  const uint32_t *sub;   // nslices * (npanels + 1): first stored element of (slice, panel); [.. + npanels] = the slice's end
  const uint16_t *rid;   // per stored element: row - (slice << rows_log2) + 1 (0: no element -- a load past the end)
  uint32_t nslices, npanels, rows_log2;
  uint32_t *pace;        // a progress board per XCD (as SweepLayout::pace)
  uint32_t lag;          // 0: no pacing
};

#ifndef ABFT_CFG_SLICE_K
#define ABFT_CFG_SLICE_K 2  // chunks of 64 elements a wave of the slice kernel keeps in flight
#endif
#ifndef ABFT_CFG_SLICE_WAVES
#define ABFT_CFG_SLICE_WAVES 8  // waves per SIMD the slice kernel is compiled for (register budget)
#endif

#ifndef ABFT_CFG_PANEL_RPT
#define ABFT_CFG_PANEL_RPT 8  // outputs per thread of the panel kernels (4: equal on config 4, -14% on config 5)
#endif
constexpr int ABFT_PANEL_ROWS_PER_THREAD = ABFT_CFG_PANEL_RPT;
constexpr int ABFT_PANEL_ROWS = 256 * ABFT_PANEL_ROWS_PER_THREAD;

// COO only: elements whose stored column no longer names the output group they are
// stored in (a silently corrupted column: modes none/constraints, or multi-bit damage
// an ECC mode cannot see).  The reference scatters such a product into
// result[corrupted col] (COO/CPUContext.cpp:120); the SpMV kernels leave it out of the
// group's sum and queue it here, and coo_fixup_kernel, launched behind every COO SpMV,
// rebuilds each receiving output in the caller's element order.  Empty on clean data.
struct MovedEntry {
  uint32_t orig;  // caller's (local) element index: the position in the reference's serial loop
  uint32_t col;   // the output the reference adds this product to (< n_out)
  double prod;    // value * vec[row], as the SpMV formed it
};
struct MovedList {
  MovedEntry *buf;     // 2 * cap entries: [0, cap) filled by the SpMV, [cap, 2 cap) sorted by the fix-up
  uint32_t *count;     // entries pushed by the SpMV in flight (reset by the fix-up)
  uint32_t cap;
};

// COO matrix: 16-byte elements {col,row,value} (COO/ecc.h:11-16) stored grouped
// by output index (col) and, inside a group, in the caller's order -- so each
// output is still summed in the reference's storage order.
struct CooDev {
  uint4 *elems;
  const uint32_t *grp_ptr;      // n_out + 1
  const uint4 *blk;             // nblk : {first group, end group, first element, end element}
  const uint32_t *orig_index;   // stored position -> caller's element index (cold path)
  const uint32_t *pos_of_orig;  // caller's element index -> stored position
  const uint2 *as_created;      // constraints mode: {col,row} of every stored element as create_matrix left it (or its complement: see kernels.hip)
  uint32_t nblk, n_out, n_in, nnz, index_base;
  MovedList moved;
  const uint32_t *gidx;         // column-block shards: global (caller's) index of local element k; else index_base + k
};

#define ABFT_COLMASK_HOST 0x00FFFFFFu

// Tuning knobs (overridable with -D for A/B builds; defaults are the measured best)
#ifndef ABFT_CFG_CSR_EPT
#define ABFT_CFG_CSR_EPT 4
#endif
#ifndef ABFT_CFG_COO_EPT
#define ABFT_CFG_COO_EPT 4
#endif
#ifndef ABFT_CFG_PANEL_EPT
#define ABFT_CFG_PANEL_EPT 8  // elements per thread per tile of the panel-layout kernel (4: -9%)
#endif
#ifndef ABFT_CFG_SWEEP_EPT
#define ABFT_CFG_SWEEP_EPT 8  // CSR elements per thread per tile of the sweep kernel
#endif
#ifndef ABFT_CFG_COO_PANEL_EPT
#define ABFT_CFG_COO_PANEL_EPT 4  // 16-byte elements per thread per tile of the COO panel kernel
#endif
#ifndef ABFT_CFG_DEAD_NT
#define ABFT_CFG_DEAD_NT 0  // non-temporal loads of values read for the last time this iteration: bit 0 w in calc_r, 1 r in calc_px, 2 p in calc_px
#endif
#ifndef ABFT_CFG_SWEEP_COUNTS_NT
#define ABFT_CFG_SWEEP_COUNTS_NT 0  // sweep kernel: the per-(segment, row) counts by non-temporal loads
#endif
#ifndef ABFT_CFG_PX_OUT_NT
#define ABFT_CFG_PX_OUT_NT 0  // calc_px writing into shadow buffers (speculation): non-temporal stores, bit 0 x, bit 1 p
#endif
#ifndef ABFT_CFG_X_NT
#define ABFT_CFG_X_NT 0  // calc_px: x (touched once per iteration) by non-temporal loads and stores
#endif
#ifndef ABFT_CFG_COO_LEAN_WAVES
#define ABFT_CFG_COO_LEAN_WAVES 8  // register budget of spmv_coo_lean_kernel, in waves per SIMD (8: 64 VGPRs)
#endif
#ifndef ABFT_CFG_COO_PANEL_XPF_AHEAD
#define ABFT_CFG_COO_PANEL_XPF_AHEAD 1  // the x prefetch of the COO panel kernel runs this many panels ahead
#endif
#ifndef ABFT_CFG_COO_PANEL_PREFETCH
#define ABFT_CFG_COO_PANEL_PREFETCH 0  // COO panel kernel: the next tile's streaming loads issued behind this tile's gathers
#endif
#ifndef ABFT_CFG_COO_SCHED_BARRIER
#define ABFT_CFG_COO_SCHED_BARRIER 1  // bit m: COO kernels of mode m keep their streaming loads together (kernels.hip)
#endif
#ifndef ABFT_CFG_COO_PANEL_SHORT_SUMS
#define ABFT_CFG_COO_PANEL_SHORT_SUMS 1  // COO panel kernel's ordered sums two-wide instead of four-wide
#endif
#ifndef ABFT_CFG_SWEEP_SHORT_SUMS
#define ABFT_CFG_SWEEP_SHORT_SUMS 1  // sweep kernel's row sums: two LDS reads in flight per step instead of four
#endif
#ifndef ABFT_CFG_SWEEP_PREFETCH
#define ABFT_CFG_SWEEP_PREFETCH 0  // bit 0: 8 rows per thread, bit 1: 16 (see spmv_sweep_kernel)
#endif
#ifndef ABFT_CFG_SCHED_BARRIER
#define ABFT_CFG_SCHED_BARRIER 1
#endif
#ifndef ABFT_CFG_UNIFORM_ROWS
#define ABFT_CFG_UNIFORM_ROWS 1  // row blocks of equal-length rows are flagged and skip the row-pointer loads
#endif
#ifndef ABFT_CFG_NT
#define ABFT_CFG_NT 1  // stream cols/vals with the non-temporal hint
#endif
#ifndef ABFT_CFG_XCD
#define ABFT_CFG_XCD 1  // tile order: 0 dispatch order, 1 contiguous range per XCD, 2 chunked
#endif
#ifndef ABFT_CFG_XCD_CHUNK
#define ABFT_CFG_XCD_CHUNK 32
#endif

constexpr int ABFT_BLOCK = 256;
constexpr int ABFT_CSR_EPT = ABFT_CFG_CSR_EPT;           // elements per thread per tile
constexpr int ABFT_CSR_TILE = ABFT_BLOCK * ABFT_CSR_EPT;  // nnz staged per block
constexpr int ABFT_COO_EPT = ABFT_CFG_COO_EPT;
constexpr int ABFT_COO_TILE = ABFT_BLOCK * ABFT_COO_EPT;
#ifndef ABFT_CFG_MAX_PARTIALS
#define ABFT_CFG_MAX_PARTIALS 2048  // 256 CUs x 8 (1024: calc_px 7 % slower; 4096: calc_r 10 % slower)
#endif
constexpr int ABFT_MAX_PARTIALS = ABFT_CFG_MAX_PARTIALS;  // reduction blocks
constexpr int ABFT_TICKET_GROUP = 32;    // blocks per first-level arrival counter
constexpr int ABFT_TICKET_WORDS = 1 + ABFT_MAX_PARTIALS / ABFT_TICKET_GROUP;  // [0] top, [1..] groups

hipError_t launch_encode_csr(int mode, uint32_t *cols, double *vals, uint32_t nnz, hipStream_t s);
hipError_t launch_encode_coo(int mode, uint4 *elems, uint32_t nnz, hipStream_t s);
hipError_t launch_inject_csr(double *vals, uint32_t *cols, const uint32_t *pos_of_orig, uint32_t index,
                             const int *bits_dev, int nbits, hipStream_t s);
hipError_t launch_inject_coo(uint4 *elems, const uint32_t *pos_of_orig, uint32_t index,
                             const int *bits_dev, int nbits, hipStream_t s);

// Pinned, device-visible result slot: the reduction's last block writes the
// scalar and the queued-event count, then publishes `seq` with a system-scope
// release store; the host spins on `seq` instead of synchronising the stream.
struct HostSlot {
  double value;
  uint32_t evcount;
  uint32_t seq;
};

// peer board (all-reduce of a pair across the processes of one node, see kernels.hip):
// [2 rows][ABFT_PEER_MAX_RANKS] slots, then one failure flag per rank
#define ABFT_PEER_MAX_RANKS 64
struct PeerSlot {
  unsigned long long v0, v1, seq, pad;
};
#define ABFT_PEER_BOARD_BYTES (2 * ABFT_PEER_MAX_RANKS * sizeof(PeerSlot) + ABFT_PEER_MAX_RANKS * sizeof(uint32_t))
// ... as the tail of a reduction: the block that finishes the shard's sum also takes it over the
// board (size == 0: no)
struct PeerArgs {
  PeerSlot *const *boards;  // replicated board (device memory): every rank's copy, by rank (device table); else NULL
  PeerSlot *board;
  unsigned long long *counter;
  uint32_t *fail;
  int rank, size;
  unsigned long long timeout_ticks;
};

// Where a reduction delivers its result.  Every block writes one partial, takes
// a ticket (agent-scope release before, acquire after for the last arriver), and
// the block that draws the last ticket adds the partials in a fixed order, so the
// value does not depend on which block that is.
struct ReduceOut {
  double *partials;          // ABFT_MAX_PARTIALS doubles
  uint32_t *ticket;          // ABFT_TICKET_WORDS counters, zero between launches (last arrivers reset them)
  double *dev_out;           // optional: {sum, queued events} as two doubles
  HostSlot *host;            // optional: device alias of the pinned slot
  const uint32_t *ev_count;  // the context's device event counter
  uint32_t seq;              // value to publish in host->seq
  PeerArgs peers;            // dev_out form: {sum, events} summed over the ranks of the board before it is stored
};

// Cross-call fusion: an SpMV on a square matrix also forms
// sum_row vec[row] * result[row], so the dot(p, w) the CG loop asks for right
// after spmv(A, p, w) needs no pass over the vectors.  Each block leaves one
// partial; a small kernel behind the SpMV (one workgroup, or several for many partials:
// launch_fuse_finalize) folds them in a fixed order and
// publishes the scalar like a reduction does.  `partials` belongs to the matrix.
struct FuseOut {
  double *partials;  // one per SpMV workgroup
  HostSlot *host;    // publish to the pinned slot (host-scalar form) ...
  double *dev_out;   // ... or {sum, queued events} to device memory (device-scalar form)
  const uint32_t *ev_count;
  uint32_t seq;
  uint32_t x_off;    // the product uses vec[x_off + row] (a shard's slot in the gathered vector)
  PeerArgs peers;    // as in ReduceOut
};

// Which row blocks (tiles) of a streaming-layout CSR matrix one launch covers:
// workgroup b takes tile first + b, plus `skip` once b >= cut.  Whole matrix:
// {0, nblk, 0, nblk}; interior rows of a shard: {t_lo, n, 0, n}; the rest:
// {0, t_lo, t_hi - t_lo, nblk - (t_hi - t_lo)}.
struct TileSpan {
  uint32_t first, cut, skip, count;
};

// sweep-layout SpMV (modes other than constraints): panels [c0, c1) in one persistent launch of
// `grid` workgroups (all resident); c0 > 0 resumes from the sums a previous launch left in y
hipError_t launch_spmv_sweep(int mode, int rpt, const CsrDev &A, const SweepLayout &L, const double *x, double *y,
                             EventRing ev, const FuseOut *fuse, uint32_t grid, uint32_t c0, uint32_t c1, hipStream_t s);
int spmv_sweep_blocks_per_cu(int mode, int rpt);  // rpt: 2, 4, 8 or 16
// slice-layout SpMV (modes other than constraints): panels [c0, c1), `grid` workgroups of 4 waves
hipError_t launch_spmv_slice(int mode, const CsrDev &A, const SliceLayout &L, const double *x, double *y, EventRing ev,
                             const FuseOut *fuse, uint32_t grid, uint32_t c0, uint32_t c1, hipStream_t s);
int spmv_slice_blocks_per_cu(int mode, uint32_t rows_log2);

// panel-layout SpMV (modes other than constraints); `grid` = resident workgroups
hipError_t launch_spmv_csr_panels(int mode, const CsrDev &A, const CsrPanels &P, const double *x, double *y,
                                  EventRing ev, const FuseOut *fuse, uint32_t grid, uint32_t chunk,
                                  hipStream_t s);
int spmv_csr_panels_blocks_per_cu(int mode, bool fuse);
hipError_t launch_spmv_coo_panels(int mode, const CooDev &A, const CsrPanels &P, const double *x, double *y,
                                  EventRing ev, const FuseOut *fuse, uint32_t grid, uint32_t chunk,
                                  hipStream_t s);
// fuse == nullptr: plain SpMV; otherwise follow with launch_fuse_finalize
// (`big`: where a multi-workgroup fold of many partials meets; see fold_partials_kernel)
struct FixArgs;
hipError_t launch_fuse_finalize(const FuseOut &f, uint32_t nblk, const ReduceOut &big, const FixArgs *fix, hipStream_t s);
hipError_t launch_spmv_csr(int mode, const CsrDev &A, const TileSpan &span, const double *x, double *y, EventRing ev,
                           const FuseOut *fuse, hipStream_t s);
hipError_t launch_spmv_coo(int mode, const CooDev &A, const double *x, double *y, EventRing ev,
                           const FuseOut *fuse, hipStream_t s);
// behind every COO SpMV: see MovedList.  With a fused product the fix-up runs inside the fold
// (launch_fuse_finalize's `fix`), else as its own one-workgroup launch.
// where output c's own elements sit: the one group (streaming layout), or its slice of
// segment (group, panel rg) in the panel layout
struct FixLayout {
  int kind;  // 0 streaming, 1 panels
  CsrPanels P;
};
struct FixArgs {
  CooDev A;
  FixLayout F;
  const double *x;
  double *y;
  double *partial0;  // the fused product's partial 0 (corrected in place), or nullptr
  uint32_t x_off;
  int ecc;           // the stored column carries ECC bits (mask them)
  int on;            // 0: nothing to do (not a COO matrix)
};
FixArgs make_fix_args(int mode, const CooDev &A, const CsrPanels *P, const double *x, double *y, const FuseOut *fuse);
hipError_t launch_coo_fixup(const FixArgs &fx, hipStream_t s);

// cg_tail_kernel (kernels.hip): fold of the SpMV's fused partials + calc_r + calc_px in one launch
struct TailArgs {
  FuseOut f;            // the SpMV's partials, dev_out = the {p.w, events} pair, peers for its all-reduce
  uint32_t nparts;      // how many partials
  uint32_t fold_nb, fold_chunk;  // many partials (> 8192): folded in fold_nb chunks first (launch_fuse_finalize's rule); else 0
  FixArgs fx;           // COO fix-up (on = 0: none)
  ReduceOut o;          // partials = the context's block partials, dev_out = the {r.r, events} pair, peers
  const double *rr;     // this iteration's r.r (device)
  double *x, *r, *p;
  const double *w;
  int n;
  uint32_t nbv;         // reduce_blocks(n): the grid calc_r_kernel / calc_px_kernel would run on
  unsigned long long *sync;  // 32 words that never go back: counters, flags and the bases a launch leaves for the next (kernels.hip)
  unsigned long long timeout_ticks;  // wall_clock64 ticks (100 MHz) a workgroup waits at a hand-off at most
};
int spmv_coo_panels_blocks_per_cu(int mode);
int spmv_coo_pc_blocks_per_cu(int mode);
int spmv_coo_lean_blocks_per_cu(int mode);
hipError_t launch_spmv_coo_lean(int mode, const CooDev &A, const CsrPanels &P, const double *x, double *y, EventRing ev,
                                const FuseOut *fuse, uint32_t grid, uint32_t chunk, hipStream_t s);
hipError_t launch_spmv_coo_pc(int mode, const CooDev &A, const CsrPanels &P, const double *x, double *y, EventRing ev,
                              const FuseOut *fuse, uint32_t grid, uint32_t chunk, hipStream_t s);
int cg_tail_blocks_per_cu(bool vec2, bool fast, int q);  // q: virtual 256-thread blocks per workgroup (4, 2, 1)
hipError_t launch_cg_tail(const TailArgs &a, bool vec2, bool fast, int q, uint32_t grid, hipStream_t s);

int reduce_blocks(int n);
hipError_t launch_dot(const double *a, const double *b, int n, const ReduceOut &out, hipStream_t s);
// alpha = num ? *num / *den : alpha (device-resident scalars: no host round trip)
hipError_t launch_calc_xr(double *x, double *r, const double *p, const double *w, double alpha,
                          const double *num, const double *den, int n, const ReduceOut &out, hipStream_t s);
hipError_t launch_calc_p(double *p, const double *r, double beta, const double *num, const double *den, int n,
                         hipStream_t s);
hipError_t launch_calc_r(double *r, const double *w, double alpha, const double *num, const double *den,
                         double *alpha_out, int n, const ReduceOut &out, hipStream_t s, double *r_out = nullptr);
hipError_t launch_calc_px(double *p, const double *r, double *x, double beta, const double *num, const double *den,
                          double alpha, const double *alpha_ptr, int n, hipStream_t s, double *p_out = nullptr,
                          double *x_out = nullptr);
hipError_t launch_axpy(double *x, const double *p, double alpha, const double *alpha_ptr, int n, hipStream_t s);
hipError_t launch_publish_pair(const double *pair, HostSlot *host, uint32_t seq, hipStream_t s);

// window exchange over shared host memory (see kernels.hip, peer_exchange_kernel): a 4 KB header
// -- ready[rank], done[rank] sequence numbers, fail[rank] -- then per rank two outboxes (parity)
#define ABFT_PEER_MAX_PIECES 63
#define ABFT_PEER_XHDR_BYTES 4096
struct PeerPiece {
  unsigned long long box_off;  // bytes from the start of the SENDER's outbox
  unsigned int vec_off;        // doubles from the start of the gathered vector (this rank's copy)
  unsigned int count;          // doubles
  int peer;                    // out: who reads it; in: whose outbox it sits in
  int pad;
};
struct PeerExchange {  // lives in device memory
  unsigned char *const *regions;  // device-memory transport: every rank's region by rank (device table); else NULL
  unsigned char *shared;       // device alias of the mapping
  unsigned long long *counter; // sequence number of the last exchange
  int rank, size, nout, nin;
  unsigned long long box_bytes, timeout_ticks;
  PeerPiece out[ABFT_PEER_MAX_PIECES], in[ABFT_PEER_MAX_PIECES];
};
hipError_t launch_peer_exchange(const PeerExchange *X, double *full, hipStream_t s);
hipError_t launch_peer_allreduce(double *pair, const PeerArgs &P, hipStream_t s);
hipError_t launch_copy(double *dst, const double *src, int n, hipStream_t s);
hipError_t launch_stream_copy(double *dst, const double *src, size_t n, hipStream_t s);
hipError_t launch_stream_read(const double *src, size_t n, double *sink, hipStream_t s);
