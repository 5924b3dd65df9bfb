// comm_rccl.cpp -- the device collectives of comm.h on RCCL (xGMI between the GPUs
// of a node).  Built with -DABFT_WITH_RCCL and the ROCm include path; without it
// the functions below are stubs and every collective is staged through the host.
#include "comm.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifdef ABFT_WITH_RCCL
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

static void check_nccl(ncclResult_t r, const char *what)
{
  if (r == ncclSuccess)
    return;
  fflush(stdout);
  fprintf(stderr, "hip backend (rccl): %s failed: %s\n", what, ncclGetErrorString(r));
  exit(2);
}

// the communicator plus a side stream for exchanges that run beside compute
struct RcclState
{
  ncclComm_t comm;
  hipStream_t side;
  hipEvent_t ready, done;
  hipStream_t cur;  // where the exchange in progress is enqueued: `side`, or the caller's stream
  bool beside;
};

static void check_hip(hipError_t e, const char *what)
{
  if (e == hipSuccess)
    return;
  fflush(stdout);
  fprintf(stderr, "hip backend (rccl): %s failed: %s\n", what, hipGetErrorString(e));
  exit(2);
}

void* abft_rccl_init(Comm *host, int device)
{
  check_hip(hipSetDevice(device), "hipSetDevice");
  // RCCL prints its version banner on stdout at this level; stdout is the driver's transcript
  if (const char *dbg = getenv("NCCL_DEBUG"))
    if (!strcmp(dbg, "VERSION"))
      unsetenv("NCCL_DEBUG");
  ncclUniqueId id;
  if (host->rank() == 0)
    check_nccl(ncclGetUniqueId(&id), "ncclGetUniqueId");
  host->bcast(&id, sizeof(id), 0);
  RcclState *st = new RcclState();
  check_nccl(ncclCommInitRank(&st->comm, host->size(), id, host->rank()), "ncclCommInitRank");
  check_hip(hipStreamCreateWithFlags(&st->side, hipStreamNonBlocking), "hipStreamCreate");
  check_hip(hipEventCreateWithFlags(&st->ready, hipEventDisableTiming), "hipEventCreate");
  check_hip(hipEventCreateWithFlags(&st->done, hipEventDisableTiming), "hipEventCreate");
  return st;
}

void abft_rccl_destroy(void *p)
{
  RcclState *st = (RcclState *)p;
  if (!st)
    return;
  (void)hipStreamSynchronize(st->side);
  ncclCommDestroy(st->comm);
  (void)hipEventDestroy(st->ready);
  (void)hipEventDestroy(st->done);
  (void)hipStreamDestroy(st->side);
  delete st;
}

// ranks of the communicator as RCCL itself counts them (ncclCommCount), and the device it sits on
int abft_rccl_comm_count(void *p, int *device)
{
  RcclState *st = (RcclState *)p;
  int n = 0, dev = -1;
  if (!st)
    return 0;
  check_nccl(ncclCommCount(st->comm, &n), "ncclCommCount");
  check_nccl(ncclCommCuDevice(st->comm, &dev), "ncclCommCuDevice");
  if (device)
    *device = dev;
  return n;
}

void abft_rccl_allreduce_sum(void *p, double *dev, int n, void *stream)
{
  RcclState *st = (RcclState *)p;
  check_nccl(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, st->comm, (hipStream_t)stream), "ncclAllReduce");
}

void abft_rccl_exchange_begin(void *p, void *stream, bool beside)
{
  RcclState *st = (RcclState *)p;
  st->beside = beside;
  st->cur = beside ? st->side : (hipStream_t)stream;
  if (!beside)
    return;
  check_hip(hipEventRecord(st->ready, (hipStream_t)stream), "hipEventRecord");
  check_hip(hipStreamWaitEvent(st->side, st->ready, 0), "hipStreamWaitEvent");
}

void abft_rccl_exchange_finish(void *p, void *stream)
{
  RcclState *st = (RcclState *)p;
  if (!st->beside)
    return;
  check_hip(hipEventRecord(st->done, st->side), "hipEventRecord");
  check_hip(hipStreamWaitEvent((hipStream_t)stream, st->done, 0), "hipStreamWaitEvent");
}

void abft_rccl_allgather(void *p, double *full, size_t slot, int rank)
{
  RcclState *st = (RcclState *)p;
  // in place: this rank's contribution already sits in its slot of the receive buffer
  check_nccl(ncclAllGather(full + (size_t)rank * slot, full, slot, ncclDouble, st->comm, st->cur), "ncclAllGather");
}

void abft_rccl_sendrecv(void *p, const std::vector<Comm::Piece> &out, const std::vector<Comm::Piece> &in)
{
  RcclState *st = (RcclState *)p;
  check_nccl(ncclGroupStart(), "ncclGroupStart");
  for (size_t k = 0; k < out.size(); k++)
    check_nccl(ncclSend(out[k].buf, out[k].bytes / sizeof(double), ncclDouble, out[k].peer, st->comm, st->cur),
               "ncclSend");
  for (size_t k = 0; k < in.size(); k++)
    check_nccl(ncclRecv(in[k].buf, in[k].bytes / sizeof(double), ncclDouble, in[k].peer, st->comm, st->cur),
               "ncclRecv");
  check_nccl(ncclGroupEnd(), "ncclGroupEnd");
}

#else

void* abft_rccl_init(Comm *, int) { return NULL; }
void  abft_rccl_destroy(void *) {}
int   abft_rccl_comm_count(void *, int *) { return 0; }
void  abft_rccl_allreduce_sum(void *, double *, int, void *) { abort(); }
void  abft_rccl_exchange_begin(void *, void *, bool) { abort(); }
void  abft_rccl_exchange_finish(void *, void *) { abort(); }
void  abft_rccl_allgather(void *, double *, size_t, int) { abort(); }
void  abft_rccl_sendrecv(void *, const std::vector<Comm::Piece> &, const std::vector<Comm::Piece> &) { abort(); }

#endif
