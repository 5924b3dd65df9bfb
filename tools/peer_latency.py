#!/usr/bin/env python3
"""Kernel time of the two shared-host-memory collectives (abft_hip_peer_board_*, abft_hip_peer_exchange_*)
with two contexts of ONE process on one GPU standing in for two ranks -- run under
`rocprofv3 --kernel-trace --stats -- python3 tools/peer_latency.py [window_doubles] [rounds]` and read
peer_exchange_kernel / peer_allreduce_kernel in the kernel stats.  Not a multi-GPU measurement: both
"ranks" cross the same PCIe link; between two GPUs each crosses its own."""
import ctypes as C
import mmap
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abft_sparse_cg_amd as amd  # noqa: E402
from abft_sparse_cg_amd import capi  # noqa: E402


class Piece(C.Structure):
    _fields_ = [("peer", C.c_int), ("vector_offset", C.c_uint32), ("count", C.c_uint32), ("box_offset", C.c_uint64)]


def shared(nbytes):
    fd = os.memfd_create("abft_peer_latency")
    os.ftruncate(fd, nbytes)
    maps = [mmap.mmap(fd, nbytes), mmap.mmap(fd, nbytes)]
    os.close(fd)
    return maps, [C.addressof(C.c_char.from_buffer(m)) for m in maps]


def main():
    win = int(sys.argv[1]) if len(sys.argv) > 1 else 3162
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    L = capi.load()
    ctxs = [amd.HIPContext("none", "csr"), amd.HIPContext("none", "csr")]
    slot = max(4 * win, 4096)
    box = (win + 1) * 8
    xb = L.abft_hip_peer_exchange_bytes(2, box)
    keep1, xaddr = shared(xb)
    keep2, baddr = shared(L.abft_hip_peer_board_bytes())
    full, pairs = [], []
    # ABFT_PEER_LATENCY=device: the round-3 transports -- a board copy and an exchange region per rank in DEVICE
    # memory (inside one process: plain pointers; across processes they travel as IPC handles)
    device = os.environ.get("ABFT_PEER_LATENCY") == "device"
    regions, boards = (C.c_void_p * 2)(), (C.c_void_p * 2)()
    if device:
        for r, c in enumerate(ctxs):
            p = C.c_void_p()
            capi.check(L.abft_hip_peer_exchange_device_alloc(c.h, 2, box, C.byref(p)))
            regions[r] = p.value
            capi.check(L.abft_hip_peer_board_device_alloc(c.h, C.byref(p)))
            boards[r] = p.value
    for r, c in enumerate(ctxs):
        o = 1 - r
        out = (Piece * 1)(Piece(o, r * slot + (slot - win if r == 0 else 0), win, 0))
        inn = (Piece * 1)(Piece(o, o * slot + (slot - win if o == 0 else 0), win, 0))
        if device:
            capi.check(L.abft_hip_peer_exchange_attach_device(c.h, regions, r, 2, box, out, 1, inn, 1, 20.0))
            capi.check(L.abft_hip_peer_board_attach_device(c.h, boards, r, 2, 20.0))
        else:
            capi.check(L.abft_hip_peer_exchange_attach(c.h, xaddr[r], xb, r, 2, box, out, 1, inn, 1, 20.0))
            capi.check(L.abft_hip_peer_board_attach(c.h, baddr[r], L.abft_hip_peer_board_bytes(), r, 2, 20.0))
        v = c.create_vector(2 * slot)
        c.upload(v, np.random.default_rng(r).random(2 * slot))
        full.append(v)
        p = c.create_vector(2)
        c.upload(p, np.array([1.0 + r, 0.0]))
        pairs.append(p)
    import time
    for _ in range(rounds):
        for c, v in zip(ctxs, full):
            capi.check(L.abft_hip_peer_exchange(c.h, v.h))
        for c, p in zip(ctxs, pairs):
            capi.check(L.abft_hip_allreduce_pair_peers(c.h, p.device_ptr))
        if _ % 16 == 15:
            for c in ctxs:
                c.synchronize()
    for c in ctxs:
        c.synchronize()
    # without the host in between: K of them captured into one graph per context, both graphs launched
    # together, wall clock / K  (the two "ranks" then run in lockstep as two GPUs would)
    K = 100
    for what in ("exchange", "allreduce"):
        graphs = []
        for c, v, p in zip(ctxs, full, pairs):
            g = C.c_void_p()
            capi.check(L.abft_hip_graph_begin(c.h))
            for _ in range(K):
                if what == "exchange":
                    capi.check(L.abft_hip_peer_exchange(c.h, v.h))
                else:
                    capi.check(L.abft_hip_allreduce_pair_peers(c.h, p.device_ptr))
            capi.check(L.abft_hip_graph_end(c.h, C.byref(g)))
            graphs.append(g)
        best = None
        for rep in range(5):
            for c in ctxs:
                c.synchronize()
            t0 = time.perf_counter()
            for g in graphs:
                capi.check(L.abft_hip_graph_launch(g))
            for c in ctxs:
                c.synchronize()
            dt = (time.perf_counter() - t0) / K * 1e6
            best = dt if best is None else min(best, dt)
        print("peer_latency (%s memory): %s in a graph of %d: %.2f us each (best of 5, incl. the gap between kernels)"
              % ("device" if device else "host", what, K, best))
        if what == "exchange" and os.environ.get("ABFT_HIP_LIB"):  # a -DABFT_DBG_STAMPS build leaves the last exchange's phase stamps
            st = np.frombuffer(keep1[0], dtype=np.uint64, count=512)[256:256 + 16].reshape(2, 8)[:, :6].astype(np.int64)
            for r in range(2):
                d = (st[r][1:] - st[r][:-1]) * 0.01
                print("peer_latency: rank %d phases us: done-wait %.2f, copy out %.2f, stores taken %.2f, flag+wait ready %.2f, "
                      "copy in %.2f" % (r, *d))

        for g in graphs:
            L.abft_hip_graph_destroy(g)
    print("peer_latency: %d rounds, windows of %d doubles, failed flags %s" % (
        rounds, win, [L.abft_hip_peer_exchange_failed(c.h) + L.abft_hip_peer_board_failed(c.h) for c in ctxs]))
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
