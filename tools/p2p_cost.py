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

# the same window copy captured into a hipGraph and replayed
try:
    a.fill_(0.0)
    b.fill_(-1.0)
    p2p()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        p2p()
    a.copy_(torch.arange(8192, dtype=torch.float64, device="cuda"))
    g.replay()
    torch.cuda.synchronize()
    ok = bool((b[:3162] == a[:3162]).all())
    print("batch_isend_irecv in a graph: replay %s" % ("copies the data" if ok else "DID NOT copy the data"), flush=True)
    timeit("graph replay of it", g.replay)
except Exception as e:  # noqa: BLE001
    print("batch_isend_irecv in a graph failed: %s" % str(e)[:300], flush=True)
dist.destroy_process_group()
