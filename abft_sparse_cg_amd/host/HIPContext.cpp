// HIPContext.cpp -- see HIPContext.h.  Shared by cg-csr and cg-coo.
#include "HIPContext.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include <chrono>
#include <cmath>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include "comm.h"
#include "partition.h"

HIPContextBase::HIPContextBase(int format, int mode)
  : ctx_(NULL), format_(format), mode_(mode), comm_(Comm::from_env()), slot_(0), n_pad_(0), n_loc_(0),
    r0_(0), use_windows_(false), overlap_(false), pair_(NULL), pair_dev_(NULL), fused_vec_(NULL), fused_res_(NULL),
    fixed_scal_(NULL), fixed_scal_dev_(NULL), board_map_(NULL), board_bytes_(0), board_kind_("board"), xchg_kind_("board"), peers_ok_(false), fuse_allreduce_(false),
    one_node_(false), xchg_map_(NULL), xchg_bytes_(0), peer_xchg_ok_(false), has_interior_(false), fixed_beside_(false)
{
  fixed_graph_[0] = fixed_graph_[1] = NULL;
  replayed_[0] = replayed_[1] = false;
  int device = comm_ ? comm_->local_rank() : 0;
  if (const char *env = getenv("ABFT_HIP_DEVICE"))
    device = atoi(env);
  check(abft_hip_init(device, &ctx_), "abft_hip_init");
  if (comm_)
  {
    comm_->enable_device_collectives(device);
    check(abft_hip_vector_create(ctx_, 2, &pair_), "abft_hip_vector_create");
    pair_dev_ = (double *)abft_hip_vector_device_ptr(pair_);
    // every process runs the whole driver; the job's stdout is rank 0's
    if (comm_->rank() != 0 && !freopen("/dev/null", "w", stdout))
      exit(2);
    setup_peer_board();
  }
}

// One zero-filled shared-memory object of `bytes`, created by rank 0 and mapped by every rank
// (unlinked again once all have it: nothing stays behind in /dev/shm).  NULL on this rank if
// anything failed here or on rank 0; the callers agree on the outcome with all_ranks().
void *HIPContextBase::shared_region(size_t bytes)
{
  const int rank = comm_->rank();
  char name[96];
  memset(name, 0, sizeof(name));
  int fd = -1;
  if (rank == 0 && one_node_)
  {
    static int serial = 0;
    snprintf(name, sizeof(name), "/abft_cg_%ld_%d_%lld", (long)getpid(), serial++,
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0)  // (a fresh object reads as zeros)
    {
      if (fd >= 0) { close(fd); shm_unlink(name); fd = -1; }
      name[0] = 0;
    }
  }
  comm_->bcast(name, sizeof(name), 0);
  if (name[0] && rank != 0)
    fd = shm_open(name, O_RDWR, 0);
  void *map = MAP_FAILED;
  if (name[0] && fd >= 0)
    map = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (fd >= 0)
    close(fd);
  comm_->barrier();  // everybody who could has mapped it
  if (rank == 0 && name[0])
    shm_unlink(name);
  return map == MAP_FAILED ? NULL : map;
}

bool HIPContextBase::all_ranks(bool mine)
{
  double v = mine ? 1.0 : 0.0;
  comm_->allreduce_sum(&v, 1);
  return v == (double)comm_->size();
}

static double comm_timeout_seconds()
{
  double timeout = 120.0;
  if (const char *t = getenv("ABFT_COMM_TIMEOUT"))
    timeout = atof(t) > 0 ? atof(t) : timeout;
  return timeout;
}

// The two scalar all-reduces of an iteration go over a board in shared host memory when all
// ranks sit on one node (they do: one process per GPU of a node): one small kernel each, no
// collective library in the iteration's critical path.  ABFT_COMM_ALLREDUCE=rccl (or tcp) keeps
// them on the collective layer; default: the board if every rank could attach it and a few
// test sums came out right, else the collective layer -- decided together, so all ranks agree.
void HIPContextBase::setup_peer_board()
{
  const int size = comm_->size(), rank = comm_->rank();
  char host[64];
  memset(host, 0, sizeof(host));
  gethostname(host, sizeof(host) - 1);
  std::vector<char> hosts((size_t)size * sizeof(host));
  comm_->allgather(host, sizeof(host), hosts.data());
  one_node_ = size <= 64;
  for (int r = 0; r < size; r++)
    one_node_ = one_node_ && !memcmp(&hosts[(size_t)r * sizeof(host)], host, sizeof(host));
  // ranks that share this rank's GPU (tests run several on the one GPU): kernels that wait inside size their grids
  // for that share (abft_hip_set_sharers)
  {
    int device = comm_->local_rank();
    if (const char *e = getenv("ABFT_HIP_DEVICE")) device = atoi(e);
    std::vector<int> devices((size_t)size);
    comm_->allgather(&device, sizeof(device), devices.data());
    int sharers = 0;
    for (int r = 0; r < size; r++)
      sharers += devices[r] == device && !memcmp(&hosts[(size_t)r * sizeof(host)], host, sizeof(host));
    check(abft_hip_set_sharers(ctx_, sharers > 0 ? sharers : 1), "abft_hip_set_sharers");
  }
  const char *env = getenv("ABFT_COMM_ALLREDUCE");
  const bool any = !env || !strcmp(env, "auto");
  const bool want_ipc = any || !strcmp(env, "ipc"), want_host = any || !strcmp(env, "board");
  if (!want_ipc && !want_host)
    return;
  // a few test sums on this very node before the board is trusted: {rank + 1 + t, 1} summed over ranks
  auto test_sums = [&]()
  {
    bool all = true;
    for (int t = 0; t < 3 && all; t++)
    {
      check(abft_hip_write_pair(ctx_, pair_dev_, (double)(rank + 1 + t), 1.0), "abft_hip_write_pair");
      check(abft_hip_allreduce_pair_peers(ctx_, pair_dev_), "abft_hip_allreduce_pair_peers");
      double v = 0.0, e = 0.0;
      check(abft_hip_read_pair(ctx_, pair_dev_, &v, &e), "abft_hip_read_pair");
      all = all_ranks(v == 0.5 * size * (size + 1) + (double)t * size && e == (double)size);
    }
    return all;
  };
  // First choice (round 3): the board in DEVICE memory, one copy per rank, the peers' copies mapped over
  // IPC -- a rank pushes its slot into every copy (xGMI stores) and polls its own: no host memory in the
  // iteration.  Taken when every rank can map every other rank's copy and the test sums come out right.
  // (IPC handles mean something on the host that made them, and the board has 64 slots -- one_node_
  // covers both: ranks on several hosts, or more than 64 of them, go to the collective layer)
  const bool ipc_possible = one_node_;
  if (want_ipc && !ipc_possible && env && !strcmp(env, "ipc"))
  {
    fprintf(stderr, "hip backend: ABFT_COMM_ALLREDUCE=ipc needs all %d ranks on one host (at most 64)\n", size);
    exit(2);
  }
  if (want_ipc && ipc_possible)
  {
    const size_t hb = abft_hip_peer_board_ipc_handle_bytes();
    std::vector<char> mine(hb, 0), handles(hb * (size_t)size, 0);
    const bool exported = abft_hip_peer_board_ipc_export(ctx_, mine.data()) == ABFT_OK;
    comm_->allgather(mine.data(), hb, handles.data());
    bool all = all_ranks(exported);
    const bool attached = all && abft_hip_peer_board_ipc_attach(ctx_, handles.data(), rank, size, comm_timeout_seconds()) == ABFT_OK;
    all = all_ranks(attached);
    if (all)
      all = test_sums();
    if (all)
    {
      peers_ok_ = true;
      board_kind_ = "device-board";
      if (getenv("ABFT_HIP_VERBOSE") && rank == 0)
        fprintf(stderr, "hip backend: scalar all-reduces over the peer board in device memory (%d ranks, IPC, %s)\n", size, host);
      return;
    }
    abft_hip_peer_board_detach(ctx_);  // (also frees an exported copy nobody attached)
    if (env && !strcmp(env, "ipc"))
    {
      fprintf(stderr, "hip backend: ABFT_COMM_ALLREDUCE=ipc, but the device-memory board could not be set up on every rank\n");
      exit(2);
    }
  }
  if (!want_host)
    return;
  board_bytes_ = abft_hip_peer_board_bytes();
  void *map = shared_region(board_bytes_);
  const bool attached = map && abft_hip_peer_board_attach(ctx_, map, board_bytes_, rank, size, comm_timeout_seconds()) == ABFT_OK;
  bool all = all_ranks(attached);
  if (all)
    all = test_sums();
  if (!all)
  {
    if (attached)
      abft_hip_peer_board_detach(ctx_);
    if (map)
      munmap(map, board_bytes_);
    if (env && !strcmp(env, "board"))
    {
      fprintf(stderr, "hip backend: ABFT_COMM_ALLREDUCE=board, but the board could not be set up on every rank\n");
      exit(2);
    }
    return;
  }
  board_map_ = map;
  peers_ok_ = true;
  board_kind_ = "board";
  if (getenv("ABFT_HIP_VERBOSE") && rank == 0)
    fprintf(stderr, "hip backend: scalar all-reduces over the peer board (%d ranks, %s)\n", size, host);
}

// The halo windows of a banded matrix travel through shared host memory too, in one kernel per
// exchange (abft_hip_peer_exchange): decided per matrix from the windows every rank already knows
// (all_need_), so all ranks lay the outboxes out alike without talking.  ABFT_COMM_EXCHANGE=rccl
// (or tcp) keeps the collective layer; ABFT_COMM_WINDOW_BYTES caps an outbox (default 1 MiB:
// beyond that the copy through host memory costs more than a collective's latency).
void HIPContextBase::setup_peer_exchange()
{
  if (peer_xchg_ok_)
  {
    abft_hip_peer_exchange_detach(ctx_);
    if (xchg_map_)
      munmap(xchg_map_, xchg_bytes_);
    xchg_map_ = NULL;
    peer_xchg_ok_ = false;
  }
  const char *env = getenv("ABFT_COMM_EXCHANGE");
  const bool any = !env || !strcmp(env, "auto");
  const bool want_ipc = any || !strcmp(env, "ipc"), want_host = any || !strcmp(env, "board");
  if (!want_ipc && !want_host)
    return;
  const int G = comm_->size(), me = comm_->rank();
  if (!one_node_ || !use_windows_ || G < 2)
    return;
  // every sender's outbox: its windows in ascending reader order (host/partition.cpp, CPU-tested)
  std::vector<WindowPiece> wout, win;
  const size_t box = plan_outboxes(all_need_, G, slot_, me, wout, win);
  if (wout.size() > 63 || win.size() > 63)
    return;
  std::vector<abft_peer_piece> out(wout.size()), in(win.size());
  for (size_t k = 0; k < wout.size() + win.size(); k++)
  {
    const WindowPiece &w = k < wout.size() ? wout[k] : win[k - wout.size()];
    abft_peer_piece &pc = k < wout.size() ? out[k] : in[k - wout.size()];
    pc.peer = w.peer; pc.vector_offset = w.vector_offset; pc.count = w.count; pc.box_offset = w.box_offset;
  }
  long long cap = 1ll << 20;
  if (const char *c = getenv("ABFT_COMM_WINDOW_BYTES")) cap = atoll(c);
  if (box == 0 || (long long)box > cap)
    return;
  // two test exchanges (both outboxes) of a vector whose entry i of slot g holds (g + 1) * 2^26 + i + round
  auto test_exchanges = [&]()
  {
    abft_hip_vector *probe = NULL;
    check(abft_hip_vector_create(ctx_, n_pad_, &probe), "abft_hip_vector_create");
    bool right = true;
    for (int round = 0; round < 2; round++)
    {
      double *h = NULL;
      check(abft_hip_vector_map(probe, &h), "abft_hip_vector_map");
      for (int i = 0; i < n_pad_; i++) h[i] = -1.0;
      for (int i = 0; i < slot_; i++) h[(size_t)me * slot_ + i] = (double)(me + 1) * 67108864.0 + i + round;
      check(abft_hip_vector_unmap(probe, h), "abft_hip_vector_unmap");
      check(abft_hip_peer_exchange(ctx_, probe), "abft_hip_peer_exchange");
      check(abft_hip_vector_map(probe, &h), "abft_hip_vector_map");
      for (size_t k = 0; k < in.size(); k++)
        for (uint32_t i = 0; i < in[k].count; i++)
        {
          const size_t at = (size_t)in[k].vector_offset + i;
          right = right && h[at] == (double)(in[k].peer + 1) * 67108864.0 + (double)(at - (size_t)in[k].peer * slot_) + round;
        }
      check(abft_hip_vector_unmap(probe, h), "abft_hip_vector_unmap");
    }
    abft_hip_vector_destroy(probe);
    return all_ranks(right && !abft_hip_peer_exchange_failed(ctx_));
  };
  const abft_peer_piece *po = out.empty() ? NULL : out.data(), *pi = in.empty() ? NULL : in.data();
  // First choice (round 3): a region per rank in DEVICE memory, the peers' regions mapped over IPC; a rank pushes
  // its windows into the reader's region (xGMI stores between GPUs) and reads only its own memory.
  if (want_ipc)
  {
    const size_t hb = abft_hip_peer_board_ipc_handle_bytes();
    std::vector<char> mine(hb, 0), handles(hb * (size_t)G, 0);
    const bool exported = abft_hip_peer_exchange_ipc_export(ctx_, G, box, mine.data()) == ABFT_OK;
    comm_->allgather(mine.data(), hb, handles.data());
    bool all = all_ranks(exported);
    const bool attached = all && abft_hip_peer_exchange_ipc_attach(ctx_, handles.data(), me, G, box, po, (int)out.size(), pi,
                                                                   (int)in.size(), comm_timeout_seconds()) == ABFT_OK;
    all = all_ranks(attached);
    if (all)
      all = test_exchanges();
    if (all)
    {
      peer_xchg_ok_ = true;
      xchg_kind_ = "device-board";
      return;
    }
    abft_hip_peer_exchange_detach(ctx_);  // (also frees an exported region nobody attached)
    if (env && !strcmp(env, "ipc"))
    {
      fprintf(stderr, "hip backend: ABFT_COMM_EXCHANGE=ipc, but the device-memory exchange could not be set up on every rank\n");
      exit(2);
    }
  }
  if (!want_host)
    return;
  xchg_bytes_ = abft_hip_peer_exchange_bytes(G, box);
  void *map = shared_region(xchg_bytes_);
  const bool attached = map && abft_hip_peer_exchange_attach(ctx_, map, xchg_bytes_, me, G, box, po, (int)out.size(), pi,
                                                             (int)in.size(), comm_timeout_seconds()) == ABFT_OK;
  bool all = all_ranks(attached);
  if (all)
    all = test_exchanges();
  if (!all)
  {
    if (attached)
      abft_hip_peer_exchange_detach(ctx_);
    if (map)
      munmap(map, xchg_bytes_);
    if (env && !strcmp(env, "board"))
    {
      fprintf(stderr, "hip backend: ABFT_COMM_EXCHANGE=board, but the exchange could not be set up on every rank\n");
      exit(2);
    }
    return;
  }
  xchg_map_ = map;
  peer_xchg_ok_ = true;
  xchg_kind_ = "board";
}

void HIPContextBase::check_peer_board()
{
  if (peer_xchg_ok_ && abft_hip_peer_exchange_failed(ctx_))
  {
    fflush(stdout);
    fprintf(stderr, "hip backend: rank %d gave up waiting for a peer in a window exchange (shared memory)\n", comm_->rank());
    exit(2);
  }
  if (peers_ok_ && abft_hip_peer_board_failed(ctx_))
  {
    fflush(stdout);
    fprintf(stderr, "hip backend: rank %d gave up waiting for a peer in an all-reduce (peer board)\n", comm_->rank());
    exit(2);
  }
}

void HIPContextBase::device_allreduce(double *dev_pair)
{
  if (fuse_allreduce_)
    return;  // the reduction that produced the pair has summed it over the ranks already
  if (peers_ok_)
    check(abft_hip_allreduce_pair_peers(ctx_, dev_pair), "abft_hip_allreduce_pair_peers");
  else if (comm_->device_collectives())
    comm_->allreduce_sum_device(dev_pair, 2, abft_hip_get_stream(ctx_));
  else
    staged_allreduce(dev_pair);
}

HIPContextBase::~HIPContextBase()
{
  if (ctx_)
  {
    report_events(true);
    for (int k = 0; k < 2; k++)
      if (fixed_graph_[k])
        abft_hip_graph_destroy(fixed_graph_[k]);
    if (fixed_scal_)
      abft_hip_vector_destroy(fixed_scal_);
    if (pair_)
      abft_hip_vector_destroy(pair_);
    abft_hip_shutdown(ctx_);  // (detaches the peer board)
    if (board_map_)
      munmap(board_map_, board_bytes_);
    if (xchg_map_)
      munmap(xchg_map_, xchg_bytes_);
  }
  if (comm_)
  {
    comm_->barrier();
    delete comm_;
  }
}

void HIPContextBase::check(int rc, const char *what)
{
  if (rc == ABFT_OK)
    return;
  fflush(stdout);
  fprintf(stderr, "hip backend: %s failed (%d): %s\n", what, rc, abft_hip_last_error());
  exit(2);
}

// Print queued events the way the reference prints them while it runs, and
// stop like it does on a fatal one.
//
// Row-partitioned: a collective -- every rank calls it at the same point of the
// driver (forced at map_vector / destroy_matrix / the destructor, or after an
// all-reduce that summed a non-zero event count).  The ranks' events (global
// element indices) are merged in index order and cut after the first fatal one,
// which is what a single-process run prints; all ranks then stop together.
static bool event_before(const abft_event &a, const abft_event &b)
{
  return a.index != b.index ? a.index < b.index : a.kind < b.kind;
}

void HIPContextBase::report_events(bool force)
{
  if (!comm_ && !force && abft_hip_pending_events(ctx_) == 0)
    return;
  // sized to the device queue: nothing is cut off here (a run that overflows the queue
  // itself ends below with the library's message, after printing what was kept)
  static std::vector<abft_event> buffer((size_t)abft_hip_event_capacity());
  abft_event *events = buffer.data();
  int count = 0, fatal = 0;
  const int drain_rc = abft_hip_drain_events(ctx_, events, (int)buffer.size(), &count, &fatal);
  if (drain_rc != ABFT_OK && drain_rc != ABFT_ERR_RANGE)
    check(drain_rc, "abft_hip_drain_events");
  std::vector<abft_event> all(events, events + count);
  if (comm_)
  {
    std::vector<char> bytes;
    std::vector<size_t> sizes;
    comm_->allgatherv(events, (size_t)count * sizeof(abft_event), bytes, sizes);
    all.resize(bytes.size() / sizeof(abft_event));
    if (!all.empty())
      memcpy(all.data(), bytes.data(), all.size() * sizeof(abft_event));
    std::stable_sort(all.begin(), all.end(), event_before);
    fatal = 0;
    for (size_t i = 0; i < all.size() && !fatal; i++)
      if (abft_event_is_fatal(all[i].kind))
      {
        fatal = 1;
        all.resize(i + 1);
      }
  }
  char line[160];
  for (size_t i = 0; i < all.size(); i++)
  {
    abft_format_event(&all[i], line, sizeof(line));
    fputs(line, stdout);
  }
  if (drain_rc == ABFT_ERR_RANGE)
    check(drain_rc, "abft_hip_drain_events");
  if (fatal)
  {
    fflush(stdout);
    exit(1);
  }
}

// {partial sum, queued events} of this rank -> {sum over ranks, total events}:
// on the GPU stream through RCCL when that is in use, else staged through the host.
double HIPContextBase::reduce_scalar(abft_hip_vector *pair)
{
  double v[2] = {0.0, 0.0};
  (void)pair;
  double *dev = pair_dev_;
  if (peers_ok_)
    check(abft_hip_allreduce_pair_peers(ctx_, dev), "abft_hip_allreduce_pair_peers");
  else if (comm_->device_collectives())
    comm_->allreduce_sum_device(dev, 2, abft_hip_get_stream(ctx_));
  check(abft_hip_read_pair(ctx_, dev, &v[0], &v[1]), "abft_hip_read_pair");
  if (!peers_ok_ && !comm_->device_collectives())
    comm_->allreduce_sum(v, 2);
  if (std::isnan(v[0]))
    check_peer_board();
  if (v[1] > 0.0)
    report_events(true);
  return v[0];
}

cg_matrix* HIPContextBase::create_matrix(const uint32_t *columns, const uint32_t *rows,
                                         const double *values, int N, int nnz)
{
  cg_matrix *M = new cg_matrix;
  M->N = N;
  M->nnz = nnz;
  M->nnz_local = nnz;
  M->nnz_before = 0;
  M->handle = NULL;
  if (!comm_)
  {
    check(abft_hip_matrix_create_shard(ctx_, format_, mode_, columns, rows, values, N, N, nnz, 0,
                                       &M->handle),
          "abft_hip_matrix_create");
    return M;
  }

  // ---- blocks of outputs (CSR rows / COO columns) of (nearly) equal non-zero count; every rank
  // ---- computes the same cut (partition.h) ----
  const int G = comm_->size();
  if (N < G)
  {
    fprintf(stderr, "hip backend: %d ranks for a matrix of %d rows\n", G, N);
    exit(2);
  }
  std::vector<int> bounds;
  if (format_ == ABFT_FMT_CSR)
    plan_bounds_sorted(rows, nnz, N, G, bounds);
  else
  {
    std::vector<long long> per((size_t)N, 0);
    for (int i = 0; i < nnz; i++)
      if (columns[i] < (uint32_t)N) per[columns[i]]++;
    plan_bounds_counts(per, N, G, bounds);
  }
  ShardPlan plan;
  plan_shard(format_, columns, rows, values, 0, (size_t)nnz, N, bounds, comm_->rank(), plan);
  adopt_plan(M, plan);
  return M;
}

// The driver generated only this rank's row block (synthetic inputs: every rank would otherwise
// build the whole matrix, 8 x 1.7 GB of host memory and 8 x the generator time for config 4).
cg_matrix* HIPContextBase::create_matrix_rows(const uint32_t *columns, const uint32_t *rows, const double *values,
                                              int N, long long nnz_total, const long long *row_bounds,
                                              long long elem0, long long count)
{
  if (!comm_ || format_ != ABFT_FMT_CSR)
    return NULL;
  cg_matrix *M = new cg_matrix;
  M->N = N;
  M->nnz = (unsigned)nnz_total;
  M->handle = NULL;
  std::vector<int> bounds((size_t)comm_->size() + 1);
  for (int g = 0; g <= comm_->size(); g++) bounds[g] = (int)row_bounds[g];
  ShardPlan plan;
  plan_shard(ABFT_FMT_CSR, columns, rows, values, (size_t)elem0, (size_t)count, N, bounds, comm_->rank(), plan);
  adopt_plan(M, plan);
  return M;
}

// shard geometry -> device matrix, exchange pattern, overlap decision
void HIPContextBase::adopt_plan(cg_matrix *M, const ShardPlan &plan)
{
  const int G = comm_->size(), me = comm_->rank();
  bounds_ = plan.bounds;
  slot_ = plan.slot; n_pad_ = plan.n_pad; r0_ = plan.out0; n_loc_ = plan.n_loc;
  const size_t cnt = plan.lout.size();
  M->nnz_before = (unsigned)plan.first;
  M->nnz_local = (unsigned)cnt;
  M->global_index = plan.global_index;
  all_need_.assign(2 * (size_t)G * G, 0);
  comm_->allgather(plan.need.data(), sizeof(int) * plan.need.size(), all_need_.data());
  long long moved = 0;
  for (size_t k = 0; k < all_need_.size(); k += 2) moved += all_need_[k + 1] - all_need_[k];
  // windows pay when they move well under half of what the all-gather moves (banded matrices: the halo)
  use_windows_ = moved * 2 < (long long)M->N * (G - 1);
  static const uint32_t none32 = 0;
  static const double none64 = 0.0;
  const uint32_t *lout = cnt ? plan.lout.data() : &none32, *pin = cnt ? plan.pin.data() : &none32;
  const double *vals = cnt ? plan.vals.data() : &none64;
  // CSR: columns = padded gather index, rows = local output; COO: columns = local output, rows = padded gather index
  const uint32_t *c = format_ == ABFT_FMT_CSR ? pin : lout, *r = format_ == ABFT_FMT_CSR ? lout : pin;
  if (plan.global_index.empty())
    check(abft_hip_matrix_create_shard(ctx_, format_, mode_, c, r, vals, n_loc_, n_pad_, (int)cnt,
                                       (uint32_t)plan.first, &M->handle),
          "abft_hip_matrix_create_shard");
  else
    check(abft_hip_matrix_create_shard_indexed(ctx_, format_, mode_, c, r, vals, n_loc_, n_pad_, (int)cnt,
                                               plan.global_index.data(), &M->handle),
          "abft_hip_matrix_create_shard_indexed");
  // the longest run of outputs that read only this rank's own slot can be multiplied while
  // the exchange is in flight (row sums are never split: results stay bit-identical)
  const bool interior = plan.interior_hi > plan.interior_lo;
  if (interior)
    check(abft_hip_matrix_set_interior(M->handle, plan.interior_lo, plan.interior_hi), "abft_hip_matrix_set_interior");
  // ... which is worth its two stream hand-offs (~20 us) only for a long exchange; a halo of a few
  // KB goes on the compute stream in front of a single SpMV launch (ABFT_CG_OVERLAP_BYTES decides)
  long long incoming = 0;
  if (use_windows_)
    for (int g = 0; g < G; g++) incoming += plan.need[2 * g + 1] - plan.need[2 * g];
  else
    incoming = (long long)(G - 1) * slot_;
  long long least = 2ll << 20;
  if (const char *env = getenv("ABFT_CG_OVERLAP_BYTES")) least = atoll(env);
  overlap_ = interior && incoming * 8 >= least;
  has_interior_ = interior;
  setup_peer_exchange();
  if (getenv("ABFT_HIP_VERBOSE"))
    fprintf(stderr, "hip backend: rank %d of %d: %s [%d,%d), %zu non-zeros from element %zu, exchange by %s over %s, "
            "interior rows [%d,%d)%s\n", me, G, format_ == ABFT_FMT_CSR ? "rows" : "columns", r0_, r0_ + n_loc_, cnt,
            plan.first, use_windows_ ? "windows" : "all-gather",
            peer_xchg_ok_ ? (xchg_map_ ? "shared memory" : "device memory (IPC)") : comm_->device_collectives() ? "RCCL" : "TCP",
            plan.interior_lo, plan.interior_hi, overlap_ ? " beside the exchange" : "");
}

void HIPContextBase::destroy_matrix(cg_matrix *mat)
{
  report_events(true);
  check(abft_hip_matrix_destroy(mat->handle), "abft_hip_matrix_destroy");
  delete mat;
}

cg_vector* HIPContextBase::create_vector(int N)
{
  cg_vector *v = new cg_vector;
  v->N = N;
  v->handle = NULL;
  v->full = NULL;
  v->full_dev = NULL;
  v->host_full = NULL;
  if (!comm_)
  {
    check(abft_hip_vector_create(ctx_, N, &v->handle), "abft_hip_vector_create");
    return v;
  }
  if (bounds_.empty() || N != bounds_.back())
  {
    fprintf(stderr, "hip backend: row-partitioned vectors follow the matrix (create_matrix first, length %d)\n",
            bounds_.empty() ? -1 : bounds_.back());
    exit(2);
  }
  // any vector may become an spmv input: its slice lives inside a gathered buffer
  check(abft_hip_vector_create(ctx_, n_pad_, &v->full), "abft_hip_vector_create");
  check(abft_hip_vector_view(v->full, comm_->rank() * slot_, n_loc_, &v->handle), "abft_hip_vector_view");
  return v;
}

void HIPContextBase::destroy_vector(cg_vector *vec)
{
  fused_vec_ = fused_res_ = NULL;
  check(abft_hip_vector_destroy(vec->handle), "abft_hip_vector_destroy");
  if (vec->full)
    check(abft_hip_vector_destroy(vec->full), "abft_hip_vector_destroy");
  free(vec->host_full);
  delete vec;
}

double* HIPContextBase::map_vector(cg_vector *v)
{
  double *host = NULL;
  check(abft_hip_vector_map(v->handle, &host), "abft_hip_vector_map");
  if (comm_)
  {
    // the driver reads and writes whole vectors: collect the slices (rank order = row order)
    if (!v->host_full && !(v->host_full = (double *)malloc(sizeof(double) * (size_t)std::max(v->N, 1))))
      exit(2);
    std::vector<char> bytes;
    std::vector<size_t> sizes;
    comm_->allgatherv(host, sizeof(double) * (size_t)n_loc_, bytes, sizes);
    if (bytes.size() != sizeof(double) * (size_t)v->N)
    {
      fprintf(stderr, "hip backend: gathered %zu bytes for a vector of %d\n", bytes.size(), v->N);
      exit(2);
    }
    memcpy(v->host_full, bytes.data(), bytes.size());
    host = v->host_full;
  }
  report_events(true);  // map synchronises: anything an earlier spmv queued is visible now
  return host;
}

void HIPContextBase::unmap_vector(cg_vector *v, double *h)
{
  fused_vec_ = fused_res_ = NULL;
  // row-partitioned: every rank's driver filled the whole vector alike; keep this rank's rows
  check(abft_hip_vector_unmap(v->handle, comm_ ? h + r0_ : h), "abft_hip_vector_unmap");
}

void HIPContextBase::copy_vector(cg_vector *dst, const cg_vector *src)
{
  fused_vec_ = fused_res_ = NULL;
  check(abft_hip_vector_copy(dst->handle, src->handle), "abft_hip_vector_copy");
}

double HIPContextBase::dot(const cg_vector *a, const cg_vector *b)
{
  if (comm_)
  {
    // dot(p, w) right after spmv(A, p, w): the shard's p.w is already in pair_
    const bool served = fused_vec_ && ((a == fused_vec_ && b == fused_res_) || (a == fused_res_ && b == fused_vec_));
    fused_vec_ = fused_res_ = NULL;  // the all-reduce below overwrites the partial
    if (!served)
      check(abft_hip_dot_dev(ctx_, a->handle, b->handle, pair_dev_),
            "abft_hip_dot_dev");
    return reduce_scalar(pair_);
  }
  double result = 0.0;
  check(abft_hip_dot(ctx_, a->handle, b->handle, &result), "abft_hip_dot");
  report_events(false);
  return result;
}

double HIPContextBase::calc_xr(cg_vector *x, cg_vector *r, const cg_vector *p, const cg_vector *w,
                               double alpha)
{
  fused_vec_ = fused_res_ = NULL;
  if (comm_)
  {
    check(abft_hip_calc_xr_dev(ctx_, x->handle, r->handle, p->handle, w->handle, alpha,
                               pair_dev_),
          "abft_hip_calc_xr_dev");
    return reduce_scalar(pair_);
  }
  double result = 0.0;
  check(abft_hip_calc_xr(ctx_, x->handle, r->handle, p->handle, w->handle, alpha, &result),
        "abft_hip_calc_xr");
  report_events(false);
  return result;
}

void HIPContextBase::calc_p(cg_vector *p, const cg_vector *r, double beta)
{
  fused_vec_ = fused_res_ = NULL;
  check(abft_hip_calc_p(ctx_, p->handle, r->handle, beta), "abft_hip_calc_p");
}

// The exchange in front of a row-partitioned spmv: every rank's slice of `v` reaches the
// ranks that read it -- an all-gather into the slots of the gathered buffer, or just the
// windows each rank reads (the halo of a banded matrix).  With RCCL it runs on a side
// stream between begin and finish, beside whatever is enqueued in between; staged through
// the host (ABFT_COMM=tcp) it is synchronous and happens in finish, i.e. strictly after.
void HIPContextBase::exchange_begin(cg_vector *v)
{
  if (peer_xchg_ok_)
  {
    check(abft_hip_peer_exchange_begin(ctx_, v->full, fixed_beside_ ? 1 : 0), "abft_hip_peer_exchange_begin");
    return;
  }
  if (!comm_->device_collectives())
    return;
  const int G = comm_->size(), me = comm_->rank();
  if (!v->full_dev)
    v->full_dev = (double *)abft_hip_vector_device_ptr(v->full);
  comm_->device_exchange_begin(abft_hip_get_stream(ctx_), overlap_);
  if (!use_windows_)
  {
    comm_->allgather_device(v->full_dev, (size_t)slot_);
    return;
  }
  std::vector<Comm::Piece> out, in;
  for (int g = 0; g < G; g++)
  {
    if (g == me) continue;
    const int *theirs = &all_need_[2 * ((size_t)g * G + me)];  // what rank g reads of my slot
    if (theirs[1] > theirs[0])
      out.push_back(Comm::Piece{g, v->full_dev + (size_t)me * slot_ + theirs[0], sizeof(double) * (size_t)(theirs[1] - theirs[0])});
    const int *mine = &all_need_[2 * ((size_t)me * G + g)];    // what I read of rank g's slot
    if (mine[1] > mine[0])
      in.push_back(Comm::Piece{g, v->full_dev + (size_t)g * slot_ + mine[0], sizeof(double) * (size_t)(mine[1] - mine[0])});
  }
  comm_->sendrecv_device(out, in);
}

void HIPContextBase::exchange_finish(cg_vector *v)
{
  if (peer_xchg_ok_)
  {
    check(abft_hip_peer_exchange_finish(ctx_), "abft_hip_peer_exchange_finish");
    return;
  }
  if (comm_->device_collectives())
  {
    comm_->device_exchange_finish(abft_hip_get_stream(ctx_));
    return;
  }
  const int G = comm_->size(), me = comm_->rank();
  double *h = NULL;
  check(abft_hip_vector_map(v->full, &h), "abft_hip_vector_map");
  std::vector<double> all(h, h + n_pad_);
  if (!use_windows_)
    comm_->allgather(h + (size_t)me * slot_, sizeof(double) * (size_t)slot_, all.data());
  else
  {
    std::vector<Comm::Piece> out, in;
    for (int g = 0; g < G; g++)
    {
      if (g == me) continue;
      const int *theirs = &all_need_[2 * ((size_t)g * G + me)];
      if (theirs[1] > theirs[0])
        out.push_back(Comm::Piece{g, h + (size_t)me * slot_ + theirs[0], sizeof(double) * (size_t)(theirs[1] - theirs[0])});
      const int *mine = &all_need_[2 * ((size_t)me * G + g)];
      if (mine[1] > mine[0])
        in.push_back(Comm::Piece{g, all.data() + (size_t)g * slot_ + mine[0], sizeof(double) * (size_t)(mine[1] - mine[0])});
    }
    comm_->exchange(out, in);
  }
  check(abft_hip_vector_unmap(v->full, all.data()), "abft_hip_vector_unmap");
}

void HIPContextBase::spmv(const cg_matrix *mat, const cg_vector *vec, cg_vector *result)
{
  if (!comm_)
  {
    check(abft_hip_spmv(ctx_, mat->handle, vec->handle, result->handle), "abft_hip_spmv");
    return;
  }
  cg_vector *v = const_cast<cg_vector *>(vec);
  const int off = comm_->rank() * slot_;
  // rows that need nothing from the peers go beside the exchange, the others after it; the
  // pair of launches equals one (include/abft_hip.h) and also leaves this shard's
  // vec.result in pair_, for the dot that usually follows
  exchange_begin(v);
  if (overlap_)
    check(abft_hip_spmv_dot_part_dev(ctx_, mat->handle, v->full, result->handle, off, pair_dev_, ABFT_PART_INTERIOR),
          "abft_hip_spmv_dot_part_dev");
  exchange_finish(v);
  check(abft_hip_spmv_dot_part_dev(ctx_, mat->handle, v->full, result->handle, off, pair_dev_,
                                   overlap_ ? ABFT_PART_BOUNDARY : ABFT_PART_ALL),
        "abft_hip_spmv_dot_part_dev");
  fused_vec_ = vec;
  fused_res_ = result;
}

// The host half of inject_bitflip: the same 1 + num_flips rand() draws, bit
// ranges and printf lines as the reference (CSR/CPUContext.cpp:135-159,
// COO/CPUContext.cpp:123-140); the XOR itself happens on the device.
void HIPContextBase::inject_bitflip(cg_matrix *mat, BitFlipKind kind, int num_flips)
{
  if (comm_)
  {
    // rank 0 draws (and prints) like the reference; the owner of the element flips it
    std::vector<int> msg(2 + std::max(num_flips, 0), 0);
    if (comm_->rank() == 0)
    {
      msg[0] = rand() % mat->nnz;
      int first = 0, width = 96;
      if (format_ == ABFT_FMT_CSR)
      {
        if (kind == VALUE) width = 64;
        else if (kind == INDEX) { first = 64; width = 32; }
      }
      else
      {
        width = 128;
        if (kind == VALUE) { first = 64; width = 64; }
        else if (kind == INDEX) width = 64;
      }
      for (int i = 0; i < num_flips; i++)
      {
        msg[2 + i] = (rand() % width) + first;
        printf("*** flipping bit %d at index %d ***\n", msg[2 + i], msg[0]);
      }
    }
    comm_->bcast(msg.data(), sizeof(int) * msg.size(), 0);
    const unsigned index = (unsigned)msg[0];
    long long local = -1;
    if (mat->global_index.empty())
    {
      if (index >= mat->nnz_before && index < mat->nnz_before + mat->nnz_local)
        local = (long long)index - mat->nnz_before;
    }
    else
    {
      std::vector<uint32_t>::const_iterator it = std::lower_bound(mat->global_index.begin(), mat->global_index.end(), index);
      if (it != mat->global_index.end() && *it == index)
        local = it - mat->global_index.begin();
    }
    if (local >= 0)
      for (int i = 0; i < num_flips; i++)
      {
        const int bit = msg[2 + i];
        // A shard stores the gather index re-based to the slot-padded vector.  Where no code protects the
        // index word (none, constraints) a flipped bit must still name the column the one-process run and the
        // reference would read -- global column ^ bit -- so it is applied in global terms: decode, flip,
        // re-base; a column outside the matrix stays outside (n_pad + its distance past N), where the kernels
        // read 0.0 and constraints mode raises "column size".  (The ECC modes repair the stored word itself:
        // the plain flip is the right one there.  COO column blocks: README, "several GPUs".)
        if (format_ == ABFT_FMT_CSR && bit >= 64 && (mode_ == ABFT_MODE_NONE || mode_ == ABFT_MODE_CONSTRAINTS))
        {
          uint32_t w[4] = {0, 0, 0, 0};
          check(abft_hip_matrix_read_element(mat->handle, (uint32_t)local, w), "abft_hip_matrix_read_element");
          const uint32_t cur = w[2];
          const int G = comm_->size();
          long long global;  // the column this stored word stands for
          if (cur < (uint32_t)n_pad_ && (int)(cur % (uint32_t)slot_) < bounds_[cur / slot_ + 1] - bounds_[cur / slot_])
            global = (long long)bounds_[cur / slot_] + cur % (uint32_t)slot_;
          else
            global = (long long)mat->N + ((long long)cur - n_pad_);  // already outside
          long long flipped = global >= 0 && global <= 0xffffffffll ? (long long)((uint32_t)global ^ (1u << (bit - 64))) : global;
          uint32_t stored;
          if (flipped < mat->N)
          {
            int g = (int)(std::upper_bound(bounds_.begin(), bounds_.begin() + G + 1, (int)flipped) - bounds_.begin()) - 1;
            stored = (uint32_t)((long long)g * slot_ + (flipped - bounds_[g]));
          }
          else
            stored = (uint32_t)std::min<long long>(0xffffffffll, (long long)n_pad_ + (flipped - mat->N));
          std::vector<int> bits;
          for (int k = 0; k < 32; k++)
            if (((cur ^ stored) >> k) & 1u) bits.push_back(64 + k);
          if (!bits.empty())
            check(abft_hip_inject(mat->handle, (uint32_t)local, bits.data(), (int)bits.size()), "abft_hip_inject");
        }
        else
          check(abft_hip_inject(mat->handle, (uint32_t)local, &bit, 1), "abft_hip_inject");
      }
    return;
  }
  int index = rand() % mat->nnz;

  int first = 0, width;
  if (format_ == ABFT_FMT_CSR)
  {
    width = 96;                                // [0,64) value, [64,96) column
    if (kind == VALUE) width = 64;
    else if (kind == INDEX) { first = 64; width = 32; }
  }
  else
  {
    width = 128;                               // [0,64) col+row, [64,128) value
    if (kind == VALUE) { first = 64; width = 64; }
    else if (kind == INDEX) width = 64;
  }

  for (int i = 0; i < num_flips; i++)
  {
    int bit = (rand() % width) + first;
    printf("*** flipping bit %d at index %d ***\n", bit, index);
    check(abft_hip_inject(mat->handle, (uint32_t)index, &bit, 1), "abft_hip_inject");
  }
}

// ---- CGContextExt ---------------------------------------------------------------------

int HIPContextBase::ext_rank() { return comm_ ? comm_->rank() : 0; }
int HIPContextBase::ext_size() { return comm_ ? comm_->size() : 1; }

// One iteration with device-resident scalars, enqueue-only (cg.cpp:97-112): the rr of this
// iteration is pair `parity`, the new one pair 1 - parity, p.w pair 2; alpha = rr / pw and
// beta = rr_new / rr are formed inside the kernels.
void HIPContextBase::fixed_iteration(cg_matrix *A, cg_vector *x, cg_vector *r, cg_vector *p, cg_vector *w, int parity)
{
  double *cur = fixed_scal_dev_ + 2 * parity, *nxt = fixed_scal_dev_ + 2 * (1 - parity), *pw = fixed_scal_dev_ + 4;
  // Everything behind the SpMV in one launch (abft_hip_cg_iteration_dev) when no collective call has to
  // sit between its parts: one process, or the board all-reduces in the kernels' tails.  ABFT_CG_TAIL=0:
  // the three calls, as rounds 2-3 ran them (same bits).
  static const bool tail_env = !(getenv("ABFT_CG_TAIL") && !strcmp(getenv("ABFT_CG_TAIL"), "0"));
  const bool one_call = tail_env && (!comm_ || fuse_allreduce_);
  if (comm_)
  {
    const int off = comm_->rank() * slot_;
    const bool split = overlap_ || fixed_beside_;
    exchange_begin(p);
    if (split)
      check(abft_hip_spmv_dot_part_dev(ctx_, A->handle, p->full, w->handle, off, pw, ABFT_PART_INTERIOR),
            "abft_hip_spmv_dot_part_dev");
    exchange_finish(p);
    if (one_call)
    {
      check(abft_hip_cg_iteration_dev(ctx_, A->handle, p->full, off, split ? ABFT_PART_BOUNDARY : ABFT_PART_ALL, x->handle,
                                      r->handle, p->handle, w->handle, cur, pw, nxt),
            "abft_hip_cg_iteration_dev");
      return;
    }
    check(abft_hip_spmv_dot_part_dev(ctx_, A->handle, p->full, w->handle, off, pw, split ? ABFT_PART_BOUNDARY : ABFT_PART_ALL),
          "abft_hip_spmv_dot_part_dev");
    device_allreduce(pw);
  }
  else if (one_call)
  {
    check(abft_hip_cg_iteration_dev(ctx_, A->handle, p->handle, 0, ABFT_PART_ALL, x->handle, r->handle, p->handle, w->handle,
                                    cur, pw, nxt),
          "abft_hip_cg_iteration_dev");
    return;
  }
  else
    check(abft_hip_spmv_dot_dev(ctx_, A->handle, p->handle, w->handle, 0, pw), "abft_hip_spmv_dot_dev");
  check(abft_hip_calc_xr_ratio_dev(ctx_, x->handle, r->handle, p->handle, w->handle, cur, pw, nxt),
        "abft_hip_calc_xr_ratio_dev");
  if (comm_)
  {
    device_allreduce(nxt);
  }
  check(abft_hip_calc_p_ratio_dev(ctx_, p->handle, r->handle, nxt, cur), "abft_hip_calc_p_ratio_dev");
}

// {value, events} on the device summed over ranks through the host (ABFT_COMM=tcp: tests on one GPU)
void HIPContextBase::staged_allreduce(double *dev_pair)
{
  double v[2] = {0.0, 0.0};
  check(abft_hip_read_pair(ctx_, dev_pair, &v[0], &v[1]), "abft_hip_read_pair");
  comm_->allreduce_sum(v, 2);
  check(abft_hip_write_pair(ctx_, dev_pair, v[0], v[1]), "abft_hip_write_pair");
}

bool HIPContextBase::run_fixed(cg_matrix *A, cg_vector *b, cg_vector *x, cg_vector *r, cg_vector *p, cg_vector *w,
                               int warmup, int steps, int blocks, double *seconds, double *rr)
{
  fused_vec_ = fused_res_ = NULL;
  if (!fixed_scal_)
  {
    check(abft_hip_vector_create(ctx_, 6, &fixed_scal_), "abft_hip_vector_create");
    fixed_scal_dev_ = (double *)abft_hip_vector_device_ptr(fixed_scal_);
  }
  (void)x;
  if (blocks < 1) blocks = 1;
  // Replay: the iteration is captured once per parity (the rr pairs swap roles) and launched as
  // a graph -- kernels, the exchange and both all-reduces are graph nodes.  Not with host-staged
  // collectives (they synchronise).  ABFT_CG_GRAPH=0 keeps the eager enqueue.  The first two
  // iterations always run eagerly: communicators and peer connections are set up by their first
  // use, which must not happen under capture.
  bool graph = !comm_ || comm_->device_collectives() || (peers_ok_ && peer_xchg_ok_);
  if (const char *env = getenv("ABFT_CG_GRAPH")) graph = graph && strcmp(env, "0") != 0;
  // ABFT_CG_EXCHANGE_BESIDE=1: replayed, the window exchange runs on a graph branch of its own next to
  // the rows that read no window.  Off by default: a branch is two stream hand-offs at replay too, and
  // with two ranks sharing the one GPU that could be measured on it cost more than the 14 us it hides
  // (laplace5:1500,1500, 2 ranks: 268 vs 217 us per iteration)
  {
    const char *env = getenv("ABFT_CG_EXCHANGE_BESIDE");
    fixed_beside_ = graph && comm_ && peer_xchg_ok_ && has_interior_ && env && !strcmp(env, "1");
  }
  // clock reads at the start and at the end of every block's timed steps, each behind a device
  // synchronisation + a barrier across ranks
  std::vector<double> marks;
  auto mark = [&]()
  {
    check(abft_hip_synchronize(ctx_), "abft_hip_synchronize");
    if (comm_) comm_->barrier();
    marks.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count());
  };
  int done = 0;
  // Every block is the whole run again -- r = b, p = r, rr = r.r, `warmup` untimed iterations, `steps` timed
  // ones -- so that rr after the last block is rr after warmup + steps iterations whatever the number of
  // blocks (comparable across records), and a matrix on which CG converges fast is not iterated on until
  // rr underflows to 0 and alpha = 0 / 0 (configs[3]'s matrix: rr falls by ~0.44 per iteration).
  for (int blk = 0; blk < blocks; blk++)
  {
    // r = b; p = r; rr = r.r  (cg.cpp:87-91), the scalar staying on the device
    copy_vector(r, b);
    copy_vector(p, r);
    check(abft_hip_dot_dev(ctx_, r->handle, r->handle, fixed_scal_dev_), "abft_hip_dot_dev");
    if (comm_ && blk == 0)
    {
      device_allreduce(fixed_scal_dev_);
      // from here on the two all-reduces of an iteration ride in the tails of the kernels that
      // finish the shard's sums (the fold of p.w, calc_xr's last block): two launches less per iteration
      // (and a restarting block's r.r arrives summed over the ranks by its own tail)
      const char *env = getenv("ABFT_COMM_FUSE_ALLREDUCE");
      if (peers_ok_ && !(env && !strcmp(env, "0")))
      {
        check(abft_hip_peer_board_fuse(ctx_, 1), "abft_hip_peer_board_fuse");
        fuse_allreduce_ = true;
      }
    }
    else if (comm_)
      device_allreduce(fixed_scal_dev_);  // (nothing to do when the reduction's tail has summed it)
    done = 0;
    for (int it = 0; it < warmup + steps; it++)
    {
      if (it == warmup)
        mark();
      const int parity = it & 1;
      if (graph && (it >= 2 || blk > 0))
      {
        if (!fixed_graph_[parity])
        {
          int rc = abft_hip_graph_begin(ctx_);
          if (rc == ABFT_OK)
          {
            fixed_iteration(A, x, r, p, w, parity);
            rc = abft_hip_graph_end(ctx_, &fixed_graph_[parity]);
          }
          if (rc != ABFT_OK)
          {
            fprintf(stderr, "hip backend: hipGraph capture of the CG iteration failed (%s); running eagerly\n",
                    abft_hip_last_error());
            graph = false;
            fixed_beside_ = false;
            fixed_graph_[parity] = NULL;
            fixed_iteration(A, x, r, p, w, parity);
            done++;
            continue;
          }
        }
        const bool first_replay = !replayed_[parity];
        check(abft_hip_graph_launch(fixed_graph_[parity]), "abft_hip_graph_launch");
        if (first_replay && comm_)
        {
          // a fresh capture that holds collectives: wait for its first replay under a deadline, so
          // that a stack on which they cannot run from a graph ends the job with a message
          // (ABFT_CG_GRAPH=0 avoids graphs) instead of hanging it
          replayed_[parity] = true;
          if (abft_hip_synchronize_timeout(ctx_, 180.0) != ABFT_OK)
          {
            fflush(stdout);
            fprintf(stderr, "hip backend: the first hipGraph replay of the CG iteration did not finish (%s); "
                    "rerun with ABFT_CG_GRAPH=0\n", abft_hip_last_error());
            _exit(70);
          }
        }
      }
      else
        fixed_iteration(A, x, r, p, w, parity);
      done++;
    }
    // the end of the block's timed region: device synchronisation + barrier, then the clock -- before any teardown
    mark();
  }
  std::vector<double> dts((size_t)blocks, 0.0);
  for (int k = 0; k < blocks; k++) dts[k] = marks[2 * k + 1] - marks[2 * k];
  // The graphs hold the device pointers of A, x, r, p, w, of the exchange's description and side
  // stream, and the fuse / beside decisions of THIS call: they must not outlive it (another matrix,
  // other vectors or a re-attached exchange would replay stale pointers).
  for (int k = 0; k < 2; k++)
  {
    if (fixed_graph_[k]) abft_hip_graph_destroy(fixed_graph_[k]);
    fixed_graph_[k] = NULL;
    replayed_[k] = false;
  }
  if (comm_) check_peer_board();
  if (comm_)
  {
    // a block took as long as its slowest rank
    std::vector<double> all((size_t)comm_->size() * blocks);
    comm_->allgather(dts.data(), sizeof(double) * blocks, all.data());
    for (int g = 0; g < comm_->size(); g++)
      for (int k = 0; k < blocks; k++) dts[k] = std::max(dts[k], all[(size_t)g * blocks + k]);
  }
  double v[2] = {0.0, 0.0};
  check(abft_hip_read_pair(ctx_, fixed_scal_dev_ + 2 * (done & 1), &v[0], &v[1]), "abft_hip_read_pair");
  if (comm_)
  {
    // what every rank actually used, one line per rank on rank 0's stdout: a multi-GPU run that nobody
    // watches must say which transports carried it (bench.py copies these lines into its record)
    char mine[192], dev_s[16] = "?";
    int dev = -1;
    const int count = comm_->rccl_comm_count(&dev);
    if (dev >= 0) snprintf(dev_s, sizeof(dev_s), "%d", dev);
    else if (const char *lr = getenv("LOCAL_RANK")) snprintf(dev_s, sizeof(dev_s), "%s", lr);
    snprintf(mine, sizeof(mine), "rank %d device %s allreduce %s exchange %s-over-%s graph %d ncclCommCount %d",
             comm_->rank(), dev_s,
             peers_ok_ ? (std::string(board_kind_) + (fuse_allreduce_ ? "-in-kernel-tails" : "")).c_str()
                       : comm_->device_collectives() ? "rccl" : "tcp",
             use_windows_ ? "windows" : "allgather",
             peer_xchg_ok_ ? xchg_kind_ : comm_->device_collectives() ? "rccl" : "tcp", graph ? 1 : 0, count);
    std::vector<char> all((size_t)comm_->size() * sizeof(mine));
    comm_->allgather(mine, sizeof(mine), all.data());
    if (comm_->rank() == 0)
      for (int k = 0; k < comm_->size(); k++)
        printf("bench_transport: %s\n", &all[(size_t)k * sizeof(mine)]);
  }
  if (getenv("ABFT_BENCH_PROFILE"))
  {
    // measurement aid, after the timed region: a few more iterations, enqueued eagerly, with every
    // SpMV launch bracketed by HIP events on the context's stream (brackets cannot be captured)
    check(abft_hip_profile_enable(ctx_, 1 << ABFT_K_SPMV), "abft_hip_profile_enable");
    check(abft_hip_profile_reset(ctx_), "abft_hip_profile_reset");
    const int extra = 2 * std::max(5, std::min(steps, 20) / 2);
    for (int it = 0; it < extra; it++)
      fixed_iteration(A, x, r, p, w, (done + it) & 1);
    double ms = 0.0;
    long launches = 0;
    check(abft_hip_profile_read(ctx_, ABFT_K_SPMV, &ms, &launches), "abft_hip_profile_read");
    check(abft_hip_profile_enable(ctx_, 0), "abft_hip_profile_enable");
    printf("bench_spmv: rank %d spmvs %d brackets %ld total_us %.3f local_rows %d local_nnz %u\n", ext_rank(), extra,
           launches, ms * 1e3, comm_ ? n_loc_ : A->N, A->nnz_local);
  }
  if (fuse_allreduce_)
  {
    check(abft_hip_peer_board_fuse(ctx_, 0), "abft_hip_peer_board_fuse");
    fuse_allreduce_ = false;
  }
  fixed_beside_ = false;
  if (comm_ || v[1] > 0.0)
    report_events(true);  // collective across ranks: every rank calls it
  if (seconds)
    for (int k = 0; k < blocks; k++) seconds[k] = dts[k];
  if (rr) *rr = v[0];
  return true;
}
