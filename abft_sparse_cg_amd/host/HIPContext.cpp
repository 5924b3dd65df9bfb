// HIPContext.cpp -- see HIPContext.h.  Shared by cg-csr and cg-coo.
#include "HIPContext.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "comm.h"

HIPContextBase::HIPContextBase(int format, int mode)
  : ctx_(NULL), format_(format), mode_(mode), comm_(Comm::from_env()), slot_(0), n_pad_(0), n_loc_(0),
    r0_(0), use_windows_(false), overlap_(false), pair_(NULL), pair_dev_(NULL), fused_vec_(NULL), fused_res_(NULL)
{
  int device = comm_ ? comm_->local_rank() : 0;
  if (const char *env = getenv("ABFT_HIP_DEVICE"))
    device = atoi(env);
  check(abft_hip_init(device, &ctx_), "abft_hip_init");
  if (comm_)
  {
    if (format_ != ABFT_FMT_CSR)
    {
      fprintf(stderr, "hip backend: the row-partitioned target shards CSR (cg-csr); cg-coo runs on one GPU\n");
      exit(2);
    }
    comm_->enable_device_collectives(device);
    check(abft_hip_vector_create(ctx_, 2, &pair_), "abft_hip_vector_create");
    pair_dev_ = (double *)abft_hip_vector_device_ptr(pair_);
    // every process runs the whole driver; the job's stdout is rank 0's
    if (comm_->rank() != 0 && !freopen("/dev/null", "w", stdout))
      exit(2);
  }
}

HIPContextBase::~HIPContextBase()
{
  if (ctx_)
  {
    report_events(true);
    if (pair_)
      abft_hip_vector_destroy(pair_);
    abft_hip_shutdown(ctx_);
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
  if (comm_->device_collectives())
    comm_->allreduce_sum_device(dev, 2, abft_hip_get_stream(ctx_));
  check(abft_hip_read_pair(ctx_, dev, &v[0], &v[1]), "abft_hip_read_pair");
  if (!comm_->device_collectives())
    comm_->allreduce_sum(v, 2);
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

  // ---- row blocks of (nearly) equal non-zero count; every rank computes the same cut ----
  const int G = comm_->size(), me = comm_->rank();
  if (N < G)
  {
    fprintf(stderr, "hip backend: %d ranks for a matrix of %d rows\n", G, N);
    exit(2);
  }
  bounds_.assign(G + 1, 0);
  bounds_[G] = N;
  for (int g = 1; g < G; g++)
  {
    int row = nnz ? (int)rows[(size_t)((unsigned long long)nnz * g / G)] : (int)((long long)N * g / G);
    row = std::max(row, bounds_[g - 1] + 1);   // at least one row each ...
    row = std::min(row, N - (G - g));          // ... and room for the ranks behind
    bounds_[g] = row;
  }
  slot_ = 0;
  for (int g = 0; g < G; g++)
    slot_ = std::max(slot_, bounds_[g + 1] - bounds_[g]);
  n_pad_ = slot_ * G;
  r0_ = bounds_[me];
  n_loc_ = bounds_[me + 1] - r0_;
  const uint32_t *lo = std::lower_bound(rows, rows + nnz, (uint32_t)r0_);
  const uint32_t *hi = std::lower_bound(rows, rows + nnz, (uint32_t)(r0_ + n_loc_));
  const size_t e0 = lo - rows, cnt = hi - lo;
  M->nnz_before = (unsigned)e0;
  M->nnz_local = (unsigned)cnt;
  // local rows; columns re-based to the slot-padded gathered vector (slot g at [g*slot, ...))
  // ... noting which window of each peer's slot this rank reads, and which rows read a peer at all
  std::vector<uint32_t> lrows(cnt ? cnt : 1), pcols(cnt ? cnt : 1);
  std::vector<int> need(2 * (size_t)G, 0);  // per peer: [lo, hi) inside its slot
  std::vector<char> remote((size_t)n_loc_, 0);
  for (int g = 0; g < G; g++) need[2 * g] = slot_;
  for (size_t i = 0; i < cnt; i++)
  {
    lrows[i] = rows[e0 + i] - (uint32_t)r0_;
    const uint32_t c = columns[e0 + i];
    if (c >= (uint32_t)N)
    {
      pcols[i] = (uint32_t)n_pad_;  // out of range stays out of range (the kernel reads 0.0, as on one GPU)
      continue;
    }
    const int owner = (int)(std::upper_bound(bounds_.begin(), bounds_.end(), (int)c) - bounds_.begin()) - 1;
    const int off = (int)c - bounds_[owner];
    pcols[i] = (uint32_t)(owner * slot_ + off);
    if (owner != me)
    {
      need[2 * owner] = std::min(need[2 * owner], off);
      need[2 * owner + 1] = std::max(need[2 * owner + 1], off + 1);
      remote[lrows[i]] = 1;
    }
  }
  for (int g = 0; g < G; g++)
    if (need[2 * g + 1] <= need[2 * g]) need[2 * g] = need[2 * g + 1] = 0;
  all_need_.assign(2 * (size_t)G * G, 0);
  comm_->allgather(need.data(), sizeof(int) * need.size(), all_need_.data());
  long long moved = 0;
  for (size_t k = 0; k < all_need_.size(); k += 2) moved += all_need_[k + 1] - all_need_[k];
  // windows pay when they move well under half of what the all-gather moves (banded matrices: the halo)
  use_windows_ = moved * 2 < (long long)N * (G - 1);
  check(abft_hip_matrix_create_shard(ctx_, format_, mode_, pcols.data(), lrows.data(), values + e0, n_loc_,
                                     n_pad_, (int)cnt, (uint32_t)e0, &M->handle),
        "abft_hip_matrix_create_shard");
  // the longest run of rows that read only this rank's own slot can be multiplied while
  // the exchange is in flight (row sums are never split: results stay bit-identical)
  int best_lo = 0, best_hi = 0, run = 0;
  for (int r = 0; r <= n_loc_; r++)
  {
    if (r < n_loc_ && !remote[r]) { run++; continue; }
    if (run > best_hi - best_lo) { best_lo = r - run; best_hi = r; }
    run = 0;
  }
  const bool interior = best_hi - best_lo >= std::max(n_loc_ / 4, 1);
  if (interior)
    check(abft_hip_matrix_set_interior(M->handle, best_lo, best_hi), "abft_hip_matrix_set_interior");
  // ... which is worth its two stream hand-offs (~20 us) only for a long exchange; a halo of a few
  // KB goes on the compute stream in front of a single SpMV launch (ABFT_CG_OVERLAP_BYTES decides)
  long long incoming = 0;
  if (use_windows_)
    for (int g = 0; g < G; g++) incoming += need[2 * g + 1] - need[2 * g];
  else
    incoming = (long long)(G - 1) * slot_;
  long long least = 2ll << 20;
  if (const char *env = getenv("ABFT_CG_OVERLAP_BYTES")) least = atoll(env);
  overlap_ = interior && incoming * 8 >= least;
  if (getenv("ABFT_HIP_VERBOSE"))
    fprintf(stderr, "hip backend: rank %d of %d: rows [%d,%d), %zu non-zeros from element %zu, exchange by %s over %s, "
            "interior rows [%d,%d)%s\n", me, G, r0_, r0_ + n_loc_, cnt, e0, use_windows_ ? "windows" : "all-gather",
            comm_->device_collectives() ? "RCCL" : "TCP", interior ? best_lo : 0, interior ? best_hi : 0,
            overlap_ ? " beside the exchange" : "");
  return M;
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
      if (kind == VALUE) width = 64;
      else if (kind == INDEX) { first = 64; width = 32; }
      for (int i = 0; i < num_flips; i++)
      {
        msg[2 + i] = (rand() % width) + first;
        printf("*** flipping bit %d at index %d ***\n", msg[2 + i], msg[0]);
      }
    }
    comm_->bcast(msg.data(), sizeof(int) * msg.size(), 0);
    const unsigned index = (unsigned)msg[0];
    if (index >= mat->nnz_before && index < mat->nnz_before + mat->nnz_local)
      for (int i = 0; i < num_flips; i++)
        check(abft_hip_inject(mat->handle, index - mat->nnz_before, &msg[2 + i], 1), "abft_hip_inject");
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
