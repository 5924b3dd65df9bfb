// HIPContext.h -- the `hip` target: a CGContext backend whose every method is
// one call into libabft_hip.so through the C ABI of include/abft_hip.h.
//
// One class template serves both executables: cg-csr instantiates it with
// ABFT_FMT_CSR (CSR/HIPContext.cpp), cg-coo with ABFT_FMT_COO
// (COO/HIPContext.cpp); each registers ("hip", mode) for the six ABFT modes
// (plus "sec" = sec7), mirroring reference CSR/CPUContext.cpp:413-421.
//
// Error behaviour is the reference's: nothing is returned; ECC / constraint
// events are printed to stdout with the reference's text, fatal ones end the
// process with exit(1) (reference CSR/CPUContext.cpp:233-234, 398-399).  Events
// raised by an asynchronous spmv are reported at the next call that returns
// data to the host (dot, calc_xr, map_vector, destroy_matrix) -- before the
// driver prints anything computed from them, so stdout order is preserved.
//
// A failing library call (no GPU, out of memory, bad shapes) is fatal too:
// message on stderr, exit(2).  There is no CPU fallback.
//
// Several GPUs: start one process per GPU with RANK / WORLD_SIZE / LOCAL_RANK /
// MASTER_ADDR / MASTER_PORT set (host/mgpu-run does; so does torch.distributed.run --no-python)
// and every process runs the unchanged driver.  The backend then keeps a row block of
// the matrix (cut by non-zeros) and the matching slices of every vector on its GPU:
// spmv is preceded by an all-gather of the input vector, dot and calc_xr end in an
// all-reduce of {partial sum, queued events}, map_vector hands the driver the whole
// vector, events carry global element indices and are printed once, by rank 0, whose
// stdout is the job's (the other ranks' is discarded).  CSR is cut by row blocks, COO by
// column blocks (its outputs).  See comm.h and partition.h.
#pragma once
#include <vector>

#include "CGContext.h"
#include "CGContextExt.h"
#include "abft_hip.h"

class Comm;
struct ShardPlan;

struct cg_matrix
{
  abft_hip_matrix *handle;
  unsigned N;
  unsigned nnz;       // of the whole matrix
  unsigned nnz_local; // sharded: elements of this rank's block; CSR: the first being element nnz_before
  unsigned nnz_before;
  std::vector<uint32_t> global_index;  // sharded COO (column blocks): the caller's index of each local element, ascending
};

struct cg_vector
{
  abft_hip_vector *handle;  // what the kernels work on: the whole vector, or this rank's slice
  int N;
  abft_hip_vector *full;    // sharded: the slot-padded gathered buffer `handle` is a view into
  double *full_dev;         // sharded: its device address, once an spmv has asked for it
  double *host_full;        // sharded: map_vector's whole-vector staging
};

class HIPContextBase : public CGContext, public CGContextExt
{
public:
  HIPContextBase(int format, int mode);
  virtual ~HIPContextBase();

  virtual cg_matrix* create_matrix(const uint32_t *columns, const uint32_t *rows,
                                   const double *values, int N, int nnz);
  virtual void destroy_matrix(cg_matrix *mat);

  virtual cg_vector* create_vector(int N);
  virtual void destroy_vector(cg_vector *vec);
  virtual double* map_vector(cg_vector *v);
  virtual void unmap_vector(cg_vector *v, double *h);
  virtual void copy_vector(cg_vector *dst, const cg_vector *src);

  virtual double dot(const cg_vector *a, const cg_vector *b);
  virtual double calc_xr(cg_vector *x, cg_vector *r, const cg_vector *p, const cg_vector *w,
                         double alpha);
  virtual void calc_p(cg_vector *p, const cg_vector *r, double beta);
  virtual void spmv(const cg_matrix *mat, const cg_vector *vec, cg_vector *result);

  virtual void inject_bitflip(cg_matrix *mat, BitFlipKind kind, int num_flips);

  // ---- CGContextExt: what this repository's driver may use beyond the reference interface ----
  virtual int ext_rank();
  virtual int ext_size();
  virtual cg_matrix* create_matrix_rows(const uint32_t *columns, const uint32_t *rows, const double *values, int N,
                                        long long nnz_total, const long long *row_bounds, long long elem0,
                                        long long count);
  virtual bool run_fixed(cg_matrix *A, cg_vector *b, cg_vector *x, cg_vector *r, cg_vector *p, cg_vector *w,
                         int warmup, int steps, int blocks, double *seconds, double *rr);

private:
  void adopt_plan(cg_matrix *M, const ShardPlan &plan);
  void staged_allreduce(double *dev_pair);
  void *shared_region(size_t bytes);       // collective: one zero-filled POSIX shared-memory object mapped by every rank
  bool all_ranks(bool mine);               // collective: true if true on every rank
  void setup_peer_exchange();              // halo windows through shared host memory (abft_hip_peer_exchange_*)
  void setup_peer_board();                 // the node-local all-reduce of the two scalars (abft_hip_peer_board_*)
  void device_allreduce(double *dev_pair); // {value, events} summed over ranks, enqueue-only where possible
  void check_peer_board();                 // an all-reduce that gave up waiting ends the job, loudly
  void fixed_iteration(cg_matrix *A, cg_vector *x, cg_vector *r, cg_vector *p, cg_vector *w, int parity);
  void check(int rc, const char *what);
  void report_events(bool force);
  double reduce_scalar(abft_hip_vector *pair);
  void exchange_begin(cg_vector *v);
  void exchange_finish(cg_vector *v);

  abft_hip_ctx *ctx_;
  int format_;
  int mode_;

  // ---- row-partitioned mode (comm_ != NULL) ----
  Comm *comm_;
  std::vector<int> bounds_;      // row-block boundaries, size() + 1 entries
  int slot_, n_pad_, n_loc_, r0_; // slot length, padded vector length, this rank's rows [r0_, r0_ + n_loc_)
  // all_need_[(src * size + dst) * 2 + {0,1}]: the window [lo, hi) of rank dst's slot that rank src reads
  std::vector<int> all_need_;
  bool use_windows_;             // exchange only those windows (banded matrices) instead of an all-gather
  bool overlap_;                 // interior rows beside the exchange (long exchanges only)
  abft_hip_vector *pair_;        // two doubles on the device: {partial sum, queued events}
  double *pair_dev_;             // ... their address (asked once: see abft_hip_vector_device_ptr)
  const cg_vector *fused_vec_;   // the last spmv also left vec.result in pair_ (until anything else runs)
  const cg_vector *fused_res_;
  // run_fixed: three {value, events} pairs on the device (rr of even / odd iterations, p.w), and the
  // captured iteration, one graph per parity
  abft_hip_vector *fixed_scal_;
  double *fixed_scal_dev_;
  abft_hip_graph *fixed_graph_[2];
  bool replayed_[2];             // the graph of that parity has completed a replay
  void *board_map_;              // shared mapping behind the peer board (NULL: not in use)
  size_t board_bytes_;
  const char *board_kind_;       // "device-board" (IPC-mapped device memory) or "board" (shared host memory)
  const char *xchg_kind_;        // the same for the window exchange
  bool peers_ok_;                // every rank attached the board and it summed correctly
  bool fuse_allreduce_;          // run_fixed: the all-reduces run in the tails of the reductions themselves
  bool one_node_;                // every rank runs on this host
  void *xchg_map_;               // shared mapping behind the window exchange (NULL: not in use)
  size_t xchg_bytes_;
  bool peer_xchg_ok_;            // the windows of this matrix travel through shared host memory
  bool has_interior_;            // the shard has rows that read no other rank's slot
  bool fixed_beside_;            // run_fixed under graph replay: the window exchange next to those rows
};

template<int FORMAT, int MODE>
class HIPContext : public HIPContextBase
{
public:
  HIPContext() : HIPContextBase(FORMAT, MODE) {}
};

#define ABFT_REGISTER_HIP_CONTEXTS(FORMAT)                                                \
  namespace                                                                               \
  {                                                                                       \
    static CGContext::Register<HIPContext<FORMAT, ABFT_MODE_NONE> >        hip_none("hip", "none");               \
    static CGContext::Register<HIPContext<FORMAT, ABFT_MODE_CONSTRAINTS> > hip_constraints("hip", "constraints"); \
    static CGContext::Register<HIPContext<FORMAT, ABFT_MODE_SED> >         hip_sed("hip", "sed");                 \
    static CGContext::Register<HIPContext<FORMAT, ABFT_MODE_SEC7> >        hip_sec7("hip", "sec7");               \
    static CGContext::Register<HIPContext<FORMAT, ABFT_MODE_SEC8> >        hip_sec8("hip", "sec8");               \
    static CGContext::Register<HIPContext<FORMAT, ABFT_MODE_SECDED> >      hip_secded("hip", "secded");           \
    static CGContext::Register<HIPContext<FORMAT, ABFT_MODE_SEC7> >        hip_sec("hip", "sec");                 \
  }
