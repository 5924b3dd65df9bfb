// comm_rccl.cpp -- the device collectives of comm.h on RCCL (xGMI between the GPUs
// of a node).  Built with -DABFT_WITH_RCCL and the ROCm include path; without it
// the functions below are stubs and every collective is staged through the host.
#include "comm.h"

#include <cstdio>
#include <cstdlib>

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

void* abft_rccl_init(Comm *host, int device)
{
  if (hipSetDevice(device) != hipSuccess)
  {
    fprintf(stderr, "hip backend (rccl): hipSetDevice(%d) failed\n", device);
    exit(2);
  }
  ncclUniqueId id;
  if (host->rank() == 0)
    check_nccl(ncclGetUniqueId(&id), "ncclGetUniqueId");
  host->bcast(&id, sizeof(id), 0);
  ncclComm_t comm;
  check_nccl(ncclCommInitRank(&comm, host->size(), id, host->rank()), "ncclCommInitRank");
  return comm;
}

void abft_rccl_destroy(void *comm)
{
  if (comm)
    ncclCommDestroy((ncclComm_t)comm);
}

void abft_rccl_allreduce_sum(void *comm, double *dev, int n, void *stream)
{
  check_nccl(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)comm, (hipStream_t)stream),
             "ncclAllReduce");
}

void abft_rccl_allgather(void *comm, double *full, size_t slot, int rank, void *stream)
{
  // in place: this rank's contribution already sits in its slot of the receive buffer
  check_nccl(ncclAllGather(full + (size_t)rank * slot, full, slot, ncclDouble, (ncclComm_t)comm,
                           (hipStream_t)stream),
             "ncclAllGather");
}

#else

void* abft_rccl_init(Comm *, int) { return NULL; }
void  abft_rccl_destroy(void *) {}
void  abft_rccl_allreduce_sum(void *, double *, int, void *) { abort(); }
void  abft_rccl_allgather(void *, double *, size_t, int, void *) { abort(); }

#endif
