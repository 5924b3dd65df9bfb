// comm.h -- the few collectives the row-partitioned `hip` target needs between
// its processes (one process per GPU, started by host/mgpu-run or any launcher
// that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT, the same
// variables torch.distributed.run sets).  The reference is single-process: no
// counterpart there.
//
// Two layers:
//   * host collectives over TCP (a star through rank 0; sums are formed in rank
//     order, so they are reproducible): rendezvous, map_vector's gather of the
//     vector slices, ECC event exchange, inject_bitflip's broadcast;
//   * device collectives on the GPU stream -- the all-reduce of {partial, events}
//     behind dot / calc_xr and the all-gather of the search vector in front of
//     spmv -- through RCCL over xGMI when built with it (comm_rccl.cpp) and
//     ABFT_COMM is not "tcp"; otherwise staged through the host layer (how several
//     ranks can share one GPU in the tests: RCCL refuses that).
#pragma once
#include <cstddef>
#include <vector>

class Comm
{
public:
  // NULL when WORLD_SIZE is unset or 1 (unless ABFT_COMM_FORCE=1)
  static Comm* from_env();
  ~Comm();

  int rank() const { return rank_; }
  int size() const { return size_; }
  int local_rank() const { return local_rank_; }

  // ---- host collectives (blocking) ----
  void bcast(void *buf, size_t bytes, int root);
  void allgather(const void *mine, size_t bytes, void *all);            // `all`: size() * bytes
  void allgatherv(const void *mine, size_t bytes, std::vector<char> &all,
                  std::vector<size_t> &sizes);                          // rank-ordered concatenation
  void allreduce_sum(double *v, int n);                                 // rank 0 adds in rank order
  void barrier();
  // point-to-point pieces, relayed by rank 0: this rank sends out[k] to rank dst[k] and
  // receives in[k] from rank src[k].  Every rank must list its pairs in ascending rank
  // order and the two sides of a pair must agree on its size.
  struct Piece { int peer; void *buf; size_t bytes; };
  void exchange(const std::vector<Piece> &out, const std::vector<Piece> &in);

  // ---- device collectives, enqueued on `stream` (a hipStream_t) ----
  // call once the process has chosen its GPU; no-op when built without RCCL or ABFT_COMM=tcp
  void enable_device_collectives(int device);
  bool device_collectives() const { return rccl_ != NULL; }
  int rccl_comm_count(int *device) const;  // ncclCommCount of the communicator (0: none) and its device
  // v[0..n) += over ranks, in place, device memory
  void allreduce_sum_device(double *dev, int n, void *stream);
  // Vector exchange: between begin and finish the two calls below run on `stream` itself, or
  // -- `beside` -- on a side stream that starts after everything enqueued on `stream` so far
  // and that finish makes `stream` wait for, so kernels enqueued on `stream` in between
  // overlap the exchange.  The two event hand-offs of that form cost ~20 us per exchange
  // (measured), so it only pays for exchanges longer than that.
  void device_exchange_begin(void *stream, bool beside);
  void device_exchange_finish(void *stream);
  // slot `rank` of `full` (size() * slot doubles, device memory) is current; fill the others
  void allgather_device(double *full, size_t slot);
  // windows: send out[k] (device memory, bytes a multiple of 8) to out[k].peer, receive in[k]
  void sendrecv_device(const std::vector<Piece> &out, const std::vector<Piece> &in);

private:
  Comm() : rank_(0), size_(1), local_rank_(0), listen_fd_(-1), rccl_(NULL) {}
  void connect_star(const char *addr, int port);
  void send_all(int fd, const void *buf, size_t n);
  void recv_all(int fd, void *buf, size_t n);

  int rank_, size_, local_rank_;
  int listen_fd_;
  std::vector<int> peers_;  // rank 0: socket of every other rank (index = rank); others: [0] = rank 0
  void *rccl_;              // ncclComm_t when RCCL is in use
};

// comm_rccl.cpp (or its stub when built without RCCL)
void* abft_rccl_init(Comm *host, int device);
void  abft_rccl_destroy(void *comm);
int   abft_rccl_comm_count(void *comm, int *device);  // ncclCommCount / ncclCommCuDevice (0: no communicator)
void  abft_rccl_allreduce_sum(void *comm, double *dev, int n, void *stream);
void  abft_rccl_exchange_begin(void *comm, void *stream, bool beside);
void  abft_rccl_exchange_finish(void *comm, void *stream);
void  abft_rccl_allgather(void *comm, double *full, size_t slot, int rank);
void  abft_rccl_sendrecv(void *comm, const std::vector<Comm::Piece> &out, const std::vector<Comm::Piece> &in);
