#!/usr/bin/env python3
"""Host-side cost of torch.distributed calls on the nccl (RCCL) backend, world size 1:
batch_isend_irecv to self (if the stack allows it), all_reduce, all_gather_into_tensor,
all_to_all_single -- what an iteration of the sharded loop pays to enqueue them."""
import os
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29741")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
a = torch.zeros(8192, dtype=torch.float64, device="cuda")
b = torch.zeros(8192, dtype=torch.float64, device="cuda")
two = torch.zeros(2, dtype=torch.float64, device="cuda")


def timeit(name, fn, n=300):
    try:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("%-28s %7.1f us per call (host)" % (name, dt / n * 1e6), flush=True)
    except Exception as e:  # noqa: BLE001
        print("%-28s failed: %s" % (name, str(e)[:200]), flush=True)


def p2p():
    ops = [dist.P2POp(dist.isend, a[:3162], 0), dist.P2POp(dist.irecv, b[:3162], 0)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()


timeit("all_reduce(2 doubles)", lambda: dist.all_reduce(two))
timeit("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(a, a[:8192]))
timeit("all_to_all_single", lambda: dist.all_to_all_single(b[:4096], a[:4096]))
timeit("batch_isend_irecv (self)", p2p)
timeit("all_to_all (lists)", lambda: dist.all_to_all([b[:3162]], [a[:3162]]))

# (A last leg used to capture the batch_isend_irecv window copy into a hipGraph.  On torch
# 2.10 / RCCL 2.26 that ended the process with a core dump -- a native abort no try/except
# here can catch; record: gpurun_out/p2p_cost.log of round 1, DESIGN.md section 5.  The
# product never captures point-to-point copies, so the leg is gone rather than guarded.)
dist.destroy_process_group()
