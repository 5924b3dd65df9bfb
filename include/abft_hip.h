/*
 * abft_hip.h -- C ABI of the MI355X (gfx950) ABFT sparse-CG engine.
 *
 * This is the drop-in boundary for the `-t hip` target of abft-sparse-cg: one
 * entry point per virtual method of the reference plugin interface
 * (reference CGContext.h:13-36), plus the few additive calls a device backend
 * needs (stream hand-over, event drain, shard creation, profiling).  Plain
 * pointers and sizes only: no C++ types, no torch types.
 *
 * Conventions
 *   - every call returns ABFT_OK (0) or a negative ABFT_ERR_* code; the text of
 *     the last failure on the calling thread is abft_hip_last_error();
 *   - handles are opaque and owned by the library until the matching destroy;
 *   - host arrays passed to *_create_* are copied before the call returns
 *     (reference cg.cpp:418-422 frees them right after create_matrix);
 *   - spmv / calc_p / copy are asynchronous on the context's HIP stream;
 *     dot / calc_xr / map / drain_events synchronise it (reference semantics:
 *     CGContext.h:27-30 return the scalar by value);
 *   - nothing here prints or exits: ECC / constraint events are queued on the
 *     device and handed to the caller by abft_hip_drain_events(), which the
 *     host-side HIPContext turns into the reference's exact printf lines and
 *     exit(1) (reference CSR/CPUContext.cpp:175-400).
 */
#ifndef ABFT_HIP_H
#define ABFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABFT_OK 0
#define ABFT_ERR_INVALID (-1)  /* bad argument / shape mismatch              */
#define ABFT_ERR_HIP (-2)      /* a HIP runtime call failed                  */
#define ABFT_ERR_NOMEM (-3)    /* host or device allocation failed           */
#define ABFT_ERR_RANGE (-4)    /* column index does not fit the ECC layout   */
#define ABFT_ERR_NODEVICE (-5) /* no usable gfx950 device                    */

/* -m modes: reference CSR/CPUContext.cpp:415-420, COO/CPUContext.cpp:383-388 */
typedef enum {
  ABFT_MODE_NONE = 0,
  ABFT_MODE_CONSTRAINTS = 1,
  ABFT_MODE_SED = 2,
  ABFT_MODE_SEC7 = 3,
  ABFT_MODE_SEC8 = 4,
  ABFT_MODE_SECDED = 5
} abft_mode;

/* storage format: cg-csr links CSR/, cg-coo links COO/ (reference Makefile:22-52) */
typedef enum { ABFT_FMT_CSR = 0, ABFT_FMT_COO = 1 } abft_format;

/* reference CGContext.h:11 */
typedef enum { ABFT_FLIP_ANY = 0, ABFT_FLIP_VALUE = 1, ABFT_FLIP_INDEX = 2 } abft_flip_kind;

/* One queued event == one printf line of the reference backend. */
typedef enum {
  ABFT_EV_SED_DETECTED = 1,     /* "[ECC] error detected at index %d"               fatal */
  ABFT_EV_CORRECTED_BIT = 2,    /* "[ECC] corrected bit %u at index %d"                   */
  ABFT_EV_CORRECTED_PARITY = 3, /* "[ECC] corrected overall parity bit at index %d"       */
  ABFT_EV_DOUBLE_BIT = 4,       /* "[ECC] double-bit error detected"                fatal */
  ABFT_EV_ROW_SIZE = 5,         /* "row size constraint violated for ..."           fatal */
  ABFT_EV_ROW_ORDER = 6,        /* "row order constraint violated ..."              fatal */
  ABFT_EV_COL_SIZE = 7,         /* "column size constraint violated ..."            fatal */
  ABFT_EV_COL_ORDER = 8,        /* "column order constraint violated ..."           fatal */
  /* not a reference line: more COO elements carry a silently corrupted column than the
   * engine's list holds (index = its capacity); results past this point are not the
   * reference's, so the event is fatal */
  ABFT_EV_MOVED_OVERFLOW = 9
} abft_event_kind;

typedef struct {
  uint32_t kind;  /* abft_event_kind                                             */
  uint32_t index; /* element index i (or row, for the CSR row events), global    */
  uint32_t bit;   /* corrected bit for ABFT_EV_CORRECTED_BIT, else 0             */
  uint32_t fmt;   /* abft_format of the matrix that raised it                    */
} abft_event;

typedef struct abft_hip_ctx abft_hip_ctx;
typedef struct abft_hip_matrix abft_hip_matrix;
typedef struct abft_hip_vector abft_hip_vector;

/* ---- library / context ------------------------------------------------- */

const char *abft_hip_last_error(void);
int abft_hip_device_count(int *count);

/* Replaces `new HIPContext` (reference CGContext::create, CGContext.cpp:9-25).
 * Binds the context to `device` and creates its stream and scratch buffers.
 * abft_hip_shutdown releases everything the context owns except its stream, which is left
 * idle in a process-wide pool for the next abft_hip_init on that device (DESIGN.md section 5,
 * "a context's stream is pooled": hipStreamDestroy per context corrupted the host heap). */
int abft_hip_init(int device, abft_hip_ctx **ctx);
int abft_hip_shutdown(abft_hip_ctx *ctx);

/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of
 * the context's own stream; NULL restores the own stream. */
int abft_hip_set_stream(abft_hip_ctx *ctx, void *hip_stream);
void *abft_hip_get_stream(abft_hip_ctx *ctx);
int abft_hip_synchronize(abft_hip_ctx *ctx);
/* The same, giving up after `seconds` (ABFT_ERR_HIP, "timed out"): for the first replay of a
 * captured graph that holds collectives, so that a stack on which they cannot run from a graph
 * ends the job with a message instead of hanging it. */
int abft_hip_synchronize_timeout(abft_hip_ctx *ctx, double seconds);

/* ---- matrix ------------------------------------------------------------ */

/* reference CSR/CPUContext.cpp:11-44 (create_matrix + generate_ecc_bits).
 * columns/rows/values are COO triplets sorted by (row, col), 0-based.
 * ECC modes need every column < 2^24 (reference CSR/CPUContext.cpp:238). */
int abft_hip_matrix_create_csr(abft_hip_ctx *ctx, int mode, const uint32_t *columns,
                               const uint32_t *rows, const double *values, int N, int nnz,
                               abft_hip_matrix **mat);
/* reference COO/CPUContext.cpp:11-36 */
int abft_hip_matrix_create_coo(abft_hip_ctx *ctx, int mode, const uint32_t *columns,
                               const uint32_t *rows, const double *values, int N, int nnz,
                               abft_hip_matrix **mat);
/* Row-block shard of a larger matrix (multi-GPU): the result vector
 * has `n_out` entries (CSR: local rows, COO: local cols, 0-based), the input
 * vector `n_in` entries (CSR: columns index it, COO: rows do); event indices
 * are reported as index_base + local element index. */
int abft_hip_matrix_create_shard(abft_hip_ctx *ctx, int format, int mode,
                                 const uint32_t *columns, const uint32_t *rows,
                                 const double *values, int n_out, int n_in, int nnz,
                                 uint32_t index_base, abft_hip_matrix **mat);
/* The same for a shard whose elements are not one contiguous run of the caller's element
 * order (COO split by column blocks: the reference's input is row-major): event indices are
 * reported as global_index[local element index] (nnz entries, ascending; NULL = as above
 * with index_base 0). */
int abft_hip_matrix_create_shard_indexed(abft_hip_ctx *ctx, int format, int mode,
                                         const uint32_t *columns, const uint32_t *rows,
                                         const double *values, int n_out, int n_in, int nnz,
                                         const uint32_t *global_index, abft_hip_matrix **mat);
/* reference CSR/CPUContext.cpp:46-52 */
int abft_hip_matrix_destroy(abft_hip_matrix *mat);
/* How the matrix is stored and run (measurement only): *layout = 0 streaming row blocks,
 * 1 column panels (chunked launches), 2 sweep (one persistent launch); *launches_per_spmv = kernel launches one abft_hip_spmv enqueues (what
 * ABFT_K_SPMV's bracket spans), without the fold of a fused product. */
int abft_hip_matrix_info(abft_hip_matrix *mat, int *layout, int *launches_per_spmv);

/* Read back the stored (ECC-encoded) arrays in the caller's element order.
 * CSR: cols[nnz], rowptr[nrows+1], values[nnz].  Any pointer may be NULL. */
int abft_hip_matrix_read_csr(abft_hip_matrix *mat, uint32_t *cols, uint32_t *rowptr,
                             double *values);
/* COO: 16-byte elements {col,row,value} (reference COO/ecc.h:11-16), nnz of them. */
int abft_hip_matrix_read_coo(abft_hip_matrix *mat, void *elements);

/* The stored words of ONE element by the caller's index: 3 for CSR {value low, value high, column word},
 * 4 for COO {column word, row, value low, value high} -- what inject flips bits of (a shard's host layer
 * reads the re-based index before it flips it in global terms). */
int abft_hip_matrix_read_element(abft_hip_matrix *mat, uint32_t index, uint32_t *words);

/* reference CSR/CPUContext.cpp:135-159 / COO/CPUContext.cpp:123-140, minus the
 * rand() draws: the host picks `index` and the bits (so the libc sequence stays
 * the reference's) and the device XORs them into element `index`.  Bit
 * numbering is the reference's: CSR 0-63 value, 64-95 column; COO 0-31 col,
 * 32-63 row, 64-127 value. */
int abft_hip_inject(abft_hip_matrix *mat, uint32_t index, const int *bits, int nbits);

/* XOR `mask` into row pointer `row` (0..N) of a CSR matrix: the array constraints mode checks
 * at reference CSR/CPUContext.cpp:173-182 and the reference's own injector never touches. */
int abft_hip_inject_rowptr(abft_hip_matrix *mat, uint32_t row, uint32_t mask);

/* ---- vectors ----------------------------------------------------------- */

/* reference CSR/CPUContext.cpp:54-75: contents are uninitialised */
int abft_hip_vector_create(abft_hip_ctx *ctx, int N, abft_hip_vector **vec);
/* A window [offset, offset+N) of `parent` (the local slice of a gathered
 * vector); does not own memory. */
int abft_hip_vector_view(abft_hip_vector *parent, int offset, int N, abft_hip_vector **vec);
int abft_hip_vector_destroy(abft_hip_vector *vec);
/* map: device -> pinned host staging, returns the host pointer (valid until
 * unmap); unmap: host staging -> device.  reference CSR/CPUContext.cpp:68-75 */
int abft_hip_vector_map(abft_hip_vector *vec, double **host);
int abft_hip_vector_unmap(abft_hip_vector *vec, double *host);
/* reference CSR/CPUContext.cpp:77-80: copies dst->N doubles */
int abft_hip_vector_copy(abft_hip_vector *dst, const abft_hip_vector *src);
/* The raw device address (stable for the life of the vector), for callers that
 * alias library memory, e.g. to hand it to a collective.  Work the caller
 * enqueues on it must be ordered against the context's stream.  A vector whose
 * address has been handed out is excluded from the deferred x update described
 * at abft_hip_calc_xr; ask once and keep the value. */
void *abft_hip_vector_device_ptr(abft_hip_vector *vec);
int abft_hip_vector_length(abft_hip_vector *vec);

/* ---- the CG kernels ---------------------------------------------------- */

/* reference CSR/CPUContext.cpp:82-90 */
int abft_hip_dot(abft_hip_ctx *ctx, const abft_hip_vector *a, const abft_hip_vector *b,
                 double *result);
/* reference CSR/CPUContext.cpp:92-105: x += alpha p; r -= alpha w; returns r.r.
 * The x half may be carried out by the abft_hip_calc_p that follows (it reads the
 * same p: one pass over p less per iteration); any other call on the context
 * applies it first, so callers observe exactly the reference's sequence.
 * ABFT_HIP_FUSE_X=0 turns this off. */
int abft_hip_calc_xr(abft_hip_ctx *ctx, abft_hip_vector *x, abft_hip_vector *r,
                     const abft_hip_vector *p, const abft_hip_vector *w, double alpha,
                     double *result);
/* reference CSR/CPUContext.cpp:107-113: p = r + beta p */
int abft_hip_calc_p(abft_hip_ctx *ctx, abft_hip_vector *p, const abft_hip_vector *r,
                    double beta);
/* reference CSR/CPUContext.cpp:115-133 and the five ABFT variants :162-411;
 * COO/CPUContext.cpp:104-121 and :142-379.  The mode is the matrix's.
 * On a square matrix the kernel also forms sum vec[row]*result[row]; an
 * abft_hip_dot(vec, result) issued before either vector changes is answered
 * from it (same API, one pass over the vectors less; ABFT_HIP_FUSE_DOT=0 turns
 * it off).  Large matrices with scattered columns are stored in a column-panel
 * layout chosen at create time (ABFT_HIP_LAYOUT=stream|panels|auto); element
 * indices seen by the caller (events, inject, read-back) are unaffected. */
int abft_hip_spmv(abft_hip_ctx *ctx, abft_hip_matrix *mat, const abft_hip_vector *vec,
                  abft_hip_vector *result);

/* Shard-local forms for the row-partitioned solver: same kernels, but the
 * result stays on the device so a collective can sum it across ranks before
 * the host reads it.  `dev_result` is a device pointer to TWO doubles:
 * [0] = the shard's partial sum, [1] = this context's queued-event count
 * (so one all-reduce also tells every rank whether any rank has events). */
int abft_hip_dot_dev(abft_hip_ctx *ctx, const abft_hip_vector *a, const abft_hip_vector *b,
                     double *dev_result);
int abft_hip_calc_xr_dev(abft_hip_ctx *ctx, abft_hip_vector *x, abft_hip_vector *r,
                         const abft_hip_vector *p, const abft_hip_vector *w, double alpha,
                         double *dev_result);
/* ... and back: the two doubles at `dev_pair` (after the collective summed them on
 * the context's stream) delivered to the host through the pinned slot it polls. */
int abft_hip_read_pair(abft_hip_ctx *ctx, const double *dev_pair, double *value, double *events);
/* ... and the other way (a sum formed on the host, e.g. by host-staged collectives). */
int abft_hip_write_pair(abft_hip_ctx *ctx, double *dev_pair, double value, double events);

/* All-reduce of such a pair across the processes of ONE node without a collective
 * library (no counterpart in the reference, which is single-process; SURVEY 8e names the
 * two scalar all-reduces per iteration).  `shared` is a page-aligned region of at least
 * abft_hip_peer_board_bytes() that every process of the job has mapped (POSIX shared
 * memory, zero-filled when created); attach registers it with the GPU.  The all-reduce is
 * one small kernel on the context's stream -- capturable -- in which each rank publishes
 * its pair on the board and adds all ranks' pairs in rank order (identical bits everywhere).
 * Every rank must enqueue the same sequence of all-reduces.  A rank that waits longer than
 * `timeout_seconds` (<= 0: 120) for a peer leaves NaN and abft_hip_peer_board_failed() = 1. */
size_t abft_hip_peer_board_bytes(void);
int abft_hip_peer_board_attach(abft_hip_ctx *ctx, void *shared, size_t bytes, int rank, int size,
                               double timeout_seconds);
int abft_hip_peer_board_detach(abft_hip_ctx *ctx);
int abft_hip_allreduce_pair_peers(abft_hip_ctx *ctx, double *dev_pair);
/* on != 0: every device-scalar reduction of the context from now on (abft_hip_dot_dev,
 * abft_hip_calc_xr_dev, abft_hip_calc_xr_ratio_dev, the product of abft_hip_spmv_dot_*_dev)
 * delivers its pair already summed over the ranks: the all-reduce runs in the tail of the
 * block that finishes the shard's sum, no kernel of its own.  All ranks switch together. */
int abft_hip_peer_board_fuse(abft_hip_ctx *ctx, int on);
int abft_hip_peer_board_failed(abft_hip_ctx *ctx);
/* The same board in DEVICE memory, one copy per rank: a rank pushes its slot into every copy (stores over
 * xGMI between the GPUs of a node) and polls only its own -- no host memory, no PCIe crossing.  Across
 * processes: every rank calls _ipc_export (allocates its copy, hands out an IPC handle of
 * abft_hip_peer_board_ipc_handle_bytes() bytes), the handles are gathered by the caller's own means, then
 * _ipc_attach maps the peers' copies (fails, leaving nothing attached, when a peer's memory cannot be reached:
 * the caller then falls back to the host-memory board above).  Inside one process (two contexts standing in
 * for two ranks): _device_alloc per rank + _attach_device with the plain pointers.  Everything else --
 * abft_hip_allreduce_pair_peers, _fuse, _failed, _detach, identical bits on every rank -- as above. */
size_t abft_hip_peer_board_ipc_handle_bytes(void);
int abft_hip_peer_board_ipc_export(abft_hip_ctx *ctx, void *handle);
int abft_hip_peer_board_ipc_attach(abft_hip_ctx *ctx, const void *handles, int rank, int size, double timeout_seconds);
int abft_hip_peer_board_device_alloc(abft_hip_ctx *ctx, void **board);
int abft_hip_peer_board_device_free(abft_hip_ctx *ctx, void *board);
int abft_hip_peer_board_attach_device(abft_hip_ctx *ctx, void *const *boards, int rank, int size, double timeout_seconds);

/* The windows of the gathered vector that a rank's peers read (the halo of a banded matrix)
 * exchanged between the processes of ONE node through shared host memory, in one capturable
 * kernel on the context's stream ("a neighbour exchange would beat an all-gather"
 * for banded matrices).  `shared`: a page-aligned, zero-filled region of at least
 * abft_hip_peer_exchange_bytes(size, outbox_bytes) mapped by every process; every rank lays
 * the windows it sends out in its outbox (`box_offset`, 8-byte aligned, the same
 * outbox_bytes on every rank; behind each window 8 more bytes belong to the library: a
 * check word) and lists the windows it receives with the offsets their senders chose.
 * `vector_offset` / `count` are in doubles from the start of the gathered vector passed to
 * abft_hip_peer_exchange.  Every rank must enqueue the same sequence of
 * exchanges.  Give-up after `timeout_seconds` (<= 0: 120): NaN in the first received window
 * and abft_hip_peer_exchange_failed() = 1. */
typedef struct {
  int peer;                 /* out: the rank that reads it; in: the rank that sends it */
  uint32_t vector_offset;   /* doubles */
  uint32_t count;           /* doubles */
  uint64_t box_offset;      /* bytes from the start of the SENDER's outbox */
} abft_peer_piece;
size_t abft_hip_peer_exchange_bytes(int size, size_t outbox_bytes);
int abft_hip_peer_exchange_attach(abft_hip_ctx *ctx, void *shared, size_t bytes, int rank, int size,
                                  size_t outbox_bytes, const abft_peer_piece *out, int nout,
                                  const abft_peer_piece *in, int nin, double timeout_seconds);
int abft_hip_peer_exchange_detach(abft_hip_ctx *ctx);
int abft_hip_peer_exchange(abft_hip_ctx *ctx, abft_hip_vector *full);
/* ... or, with beside != 0, on a side stream behind everything enqueued so far: what the caller
 * enqueues on the context's stream until _finish runs next to the exchange (an SpMV's rows that
 * read no window: abft_hip_spmv_dot_part_dev, ABFT_PART_INTERIOR); _finish makes the context's
 * stream wait for it.  (Two stream hand-offs, also when replayed from a graph: worth it only
 * for exchanges longer than those; the C++ host leaves it off) */
int abft_hip_peer_exchange_begin(abft_hip_ctx *ctx, abft_hip_vector *full, int beside);
int abft_hip_peer_exchange_finish(abft_hip_ctx *ctx);
int abft_hip_peer_exchange_failed(abft_hip_ctx *ctx);
/* The same exchange through DEVICE memory (round 3): every rank owns a region of
 * abft_hip_peer_exchange_bytes(size, outbox_bytes) in its GPU's memory, laid out like the shared one, and a rank
 * PUSHES each window -- and its sequence words -- into the reader's region (stores over xGMI between the GPUs of
 * a node); all waiting and reading is on the rank's own memory.  Across processes: _ipc_export (allocates the
 * region, hands out an IPC handle of abft_hip_peer_board_ipc_handle_bytes() bytes), the handles gathered by the
 * caller's own means, _ipc_attach (fails, leaving nothing attached, when a peer's region cannot be mapped).
 * Inside one process: _device_alloc per rank + _attach_device with the plain pointers.  Window lists, outbox
 * layout, abft_hip_peer_exchange[_begin/_finish], _failed, _detach: as above. */
int abft_hip_peer_exchange_ipc_export(abft_hip_ctx *ctx, int size, size_t outbox_bytes, void *handle);
int abft_hip_peer_exchange_ipc_attach(abft_hip_ctx *ctx, const void *handles, int rank, int size, size_t outbox_bytes,
                                      const abft_peer_piece *out, int nout, const abft_peer_piece *in, int nin,
                                      double timeout_seconds);
int abft_hip_peer_exchange_device_alloc(abft_hip_ctx *ctx, int size, size_t outbox_bytes, void **region);
int abft_hip_peer_exchange_device_free(abft_hip_ctx *ctx, void *region);
int abft_hip_peer_exchange_attach_device(abft_hip_ctx *ctx, void *const *regions, int rank, int size, size_t outbox_bytes,
                                         const abft_peer_piece *out, int nout, const abft_peer_piece *in, int nin,
                                         double timeout_seconds);

/* Device-scalar forms, for loops that keep alpha and beta on the device (no host
 * round trip per iteration; the row-partitioned solver's fixed-iteration loop):
 *   spmv_dot_dev        result = A vec, and dev_result = {sum_row vec[vec_offset+row]
 *                       * result[row], queued events}  (vec_offset: the shard's slot
 *                       in the gathered vector; 0 on one GPU)
 *   calc_xr_ratio_dev   calc_xr with alpha = *dev_num / *dev_den  (cg.cpp:102)
 *   calc_p_ratio_dev    calc_p  with beta  = *dev_num / *dev_den  (cg.cpp:109)
 * All three are asynchronous; every pointer is device memory. */
int abft_hip_spmv_dot_dev(abft_hip_ctx *ctx, abft_hip_matrix *mat, const abft_hip_vector *vec,
                          abft_hip_vector *result, int vec_offset, double *dev_result);
int abft_hip_calc_xr_ratio_dev(abft_hip_ctx *ctx, abft_hip_vector *x, abft_hip_vector *r,
                               const abft_hip_vector *p, const abft_hip_vector *w,
                               const double *dev_num, const double *dev_den, double *dev_result);
int abft_hip_calc_p_ratio_dev(abft_hip_ctx *ctx, abft_hip_vector *p, const abft_hip_vector *r,
                              const double *dev_num, const double *dev_den);

/* SpMV in two parts, so a shard can multiply the rows that need nothing from its
 * peers while the exchange of the input vector is still in flight.
 * abft_hip_matrix_set_interior declares rows [row_lo, row_hi) of the matrix
 * "interior": every input entry they read is in place before the exchange.  Then
 *   part = ABFT_PART_INTERIOR   multiplies (whole row blocks of) those rows only,
 *   part = ABFT_PART_BOUNDARY   all the others, and completes the fused product;
 * issued in that order on unchanged vectors, the two calls together equal one
 * ABFT_PART_ALL call bit for bit, events included.  Row sums are never split
 * (an output's additions keep the reference's order), so this is a split by
 * rows, not by columns.  Layouts without row-block launches (panels, COO) keep
 * the interior part empty: INTERIOR returns at once and BOUNDARY does the work. */
typedef enum { ABFT_PART_ALL = 0, ABFT_PART_INTERIOR = 1, ABFT_PART_BOUNDARY = 2 } abft_part;
int abft_hip_matrix_set_interior(abft_hip_matrix *mat, int row_lo, int row_hi);
int abft_hip_spmv_part(abft_hip_ctx *ctx, abft_hip_matrix *mat, const abft_hip_vector *vec,
                       abft_hip_vector *result, int part);
int abft_hip_spmv_dot_part_dev(abft_hip_ctx *ctx, abft_hip_matrix *mat, const abft_hip_vector *vec,
                               abft_hip_vector *result, int vec_offset, double *dev_result, int part);

/* SpMV by ranges of column panels, for a shard whose input vector arrives slot by slot (an
 * exchange pipelined with the multiplication).  A matrix in the sweep layout is stored in
 * *npanels panels of *width input entries each (abft_hip_matrix_panels; any other layout:
 * one panel) and its rows are summed panel by panel in ascending order, so
 *   spmv_dot_range_dev(.., 0, k), spmv_dot_range_dev(.., k, npanels)
 * on unchanged vectors equal one abft_hip_spmv_dot_dev bit for bit: the first range starts
 * the row sums, a later one continues from what `result` holds, the one that ends at
 * npanels completes the fused product in dev_result (NULL: plain SpMV).  Panels [c0, c1)
 * read input entries [c0 * width, c1 * width) only. */
int abft_hip_matrix_panels(abft_hip_matrix *mat, int *npanels, int *width);
int abft_hip_spmv_dot_range_dev(abft_hip_ctx *ctx, abft_hip_matrix *mat, const abft_hip_vector *vec,
                                abft_hip_vector *result, int vec_offset, double *dev_result, int c0, int c1);

/* Speculation (round 4, opt-in: ABFT_HIP_SPECULATE=1; cross-call fusion behind the unchanged host-scalar API,
 * reference loop cg.cpp:97-112): once the library has seen one iteration of the CG loop -- spmv(A,p,w), [dot(p,w)],
 * calc_xr(x,r,p,w,alpha), calc_p(p,r,beta) on vectors it alone can see -- it enqueues, right behind the next
 * spmv(A,p,w), that iteration's r half with alpha = r.r / p.w formed on the device, into a shadow buffer; a calc_xr
 * that arrives with the same vectors and a bit-identical alpha takes it over (r's buffer and the shadow are swapped, no
 * launch, the GPU did not wait for the scalar's round trip) and at once enqueues the x / p half (x in place -- it is
 * due --, p with beta = r.r_new / r.r into a shadow), which calc_p takes over likewise; anything else drops the
 * shadows and runs the call as written.  The results are the same bits either way.  This reports how often a
 * speculated iteration was taken over / dropped. */
int abft_hip_speculation_stats(abft_hip_ctx *ctx, long *taken, long *dropped);

/* `processes` processes drive this library on the context's device at the same time (ranks sharing one GPU:
 * tests).  Launches whose workgroups wait for each other inside the kernel (abft_hip_cg_iteration_dev's one-launch
 * tail) then size their grids for that share of the device, so that every process's workgroups are resident
 * together.  Default 1: one process per GPU. */
int abft_hip_set_sharers(abft_hip_ctx *ctx, int processes);

/* One CG iteration behind its exchange, scalars on the device (reference loop cg.cpp:97-112):
 *   w = A vec [part: ABFT_PART_ALL, or ABFT_PART_BOUNDARY after an ABFT_PART_INTERIOR call],
 *   dev_pw = {vec[vec_offset..] . w, events};  alpha = dev_rr[0] / dev_pw[0];
 *   r -= alpha w;  dev_rr_new = {r . r, events};  beta = dev_rr_new[0] / dev_rr[0];
 *   x += alpha p;  p = r + beta p.
 * Bit for bit what abft_hip_spmv_dot_part_dev + abft_hip_calc_xr_ratio_dev + abft_hip_calc_p_ratio_dev
 * leave behind (with abft_hip_peer_board_fuse the two scalars arrive summed over the ranks, as there),
 * but everything behind the SpMV -- the fold of the fused product, the r half, the x / p half and the
 * two board all-reduces -- is ONE launch of co-resident workgroups where that applies (vectors of at most
 * 2^22 entries -- beyond that the three kernels are faster --, x private to the library and no operand
 * aliasing another; ABFT_HIP_TAIL=0 keeps the three kernels).  p is the
 * caller's view of vec's slot (the vector the SpMV read).  Enqueue-only: capturable. */
int abft_hip_cg_iteration_dev(abft_hip_ctx *ctx, abft_hip_matrix *mat, const abft_hip_vector *vec, int vec_offset,
                              int part, abft_hip_vector *x, abft_hip_vector *r, abft_hip_vector *p,
                              abft_hip_vector *w, const double *dev_rr, double *dev_pw, double *dev_rr_new);

/* ---- graph replay ------------------------------------------------------ */

/* Capture everything enqueued on the context's stream between begin and end -- the
 * asynchronous calls of this library (spmv*, *_dev forms, calc_p*, copy) and whatever the
 * caller enqueues on abft_hip_get_stream() meanwhile (RCCL collectives) -- into a hipGraph
 * and replay it with one launch.  Nothing that synchronises may be called in between
 * (dot / calc_xr with host results, map, drain_events), kernel brackets must be off, and a
 * calc_xr must be followed by its calc_p inside the same capture.  The fixed-iteration CG
 * loop of the drivers (-c 0: cg.cpp:94-118 with alpha and beta kept on the device) is
 * captured this way. */
typedef struct abft_hip_graph abft_hip_graph;
int abft_hip_graph_begin(abft_hip_ctx *ctx);
int abft_hip_graph_end(abft_hip_ctx *ctx, abft_hip_graph **graph);
int abft_hip_graph_launch(abft_hip_graph *graph);
int abft_hip_graph_destroy(abft_hip_graph *graph);

/* ---- events ------------------------------------------------------------ */

/* Synchronise, then move the queued events to `buf` (at most `cap`), sorted
 * by (index, kind) and cut after the first fatal one -- the order a
 * single-threaded reference run prints them in.  *count = events returned,
 * *fatal = 1 if the last one is fatal (the reference would have exit(1)ed). */
int abft_hip_drain_events(abft_hip_ctx *ctx, abft_event *buf, int cap, int *count, int *fatal);
/* Capacity of the device event queue = the `cap` with which drain never truncates.
 * drain returns ABFT_ERR_RANGE (after filling buf / count / fatal) if the device queued
 * more events than that, or if `cap` is smaller than the events due. */
int abft_hip_event_capacity(void);
/* Number of events queued as of the last synchronising call (no sync). */
int abft_hip_pending_events(abft_hip_ctx *ctx);
/* The reference's exact printf line for an event, including the newline. */
int abft_format_event(const abft_event *ev, char *buf, size_t cap);
int abft_event_is_fatal(uint32_t kind);

/* ---- measurement ------------------------------------------------------- */

typedef enum {
  ABFT_K_SPMV = 0,
  ABFT_K_DOT = 1,
  ABFT_K_CALC_XR = 2,
  ABFT_K_CALC_P = 3,
  ABFT_K_COUNT = 4
} abft_kernel_id;

/* `mask` bit k (1 << abft_kernel_id) brackets every launch of kernel k with HIP
 * events on the context's stream; 0 switches profiling off.  A bracket costs
 * two event records per launch, so measure throughput with only the kernel of
 * interest enabled.  abft_hip_profile_read synchronises and returns the summed
 * device time (ms) and launch count since the last reset.  With the fused dot
 * (spmv on a square matrix), ABFT_K_DOT times the small fold kernel that is left
 * of dot(p, w).  abft_hip_profile_stride(n) brackets only every n-th launch of an
 * enabled kernel (default 1): a sampled average at 1/n of the cost. */
int abft_hip_profile_enable(abft_hip_ctx *ctx, int mask);
int abft_hip_profile_stride(abft_hip_ctx *ctx, int stride);
int abft_hip_profile_reset(abft_hip_ctx *ctx);
int abft_hip_profile_read(abft_hip_ctx *ctx, int kernel, double *total_ms, long *launches);

/* Device streaming-copy bandwidth probe (bytes moved = 2*bytes per rep):
 * the measured-peak denominator SURVEY 8(d) asks for beside the 8 TB/s spec. */
int abft_hip_stream_probe(abft_hip_ctx *ctx, size_t bytes, int reps, double *gbps_copy,
                          double *gbps_read);

#ifdef __cplusplus
}
#endif
#endif /* ABFT_HIP_H */
