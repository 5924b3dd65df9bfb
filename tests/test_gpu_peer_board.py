"""The node-local all-reduce of the two CG scalars (include/abft_hip.h: abft_hip_peer_board_*):
every rank publishes {value, events} on a board in shared host memory and adds all ranks' pairs
in rank order.  Here: two contexts of ONE process share an anonymous mapping (two streams of the
one GPU stand in for two GPUs); the C++ drivers' use of it across processes is in test_gpu_cli.py.
No counterpart in the reference (single-process; SURVEY 8e names the two all-reduces)."""
import ctypes as C
import math
import mmap
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def pair_of_contexts():
    import abft_sparse_cg_amd as amd
    from abft_sparse_cg_amd import capi
    L = capi.load()
    nbytes = L.abft_hip_peer_board_bytes()
    assert nbytes % mmap.PAGESIZE == 0
    # one shared object mapped twice, as two processes would map it (zero-filled, page-aligned)
    fd = os.memfd_create("abft_board_test")
    os.ftruncate(fd, nbytes)
    maps = [mmap.mmap(fd, nbytes), mmap.mmap(fd, nbytes)]
    os.close(fd)
    addr = [C.addressof(C.c_char.from_buffer(m)) for m in maps]
    ctxs = [amd.HIPContext("none", "csr"), amd.HIPContext("none", "csr")]
    pairs = [c.create_vector(2) for c in ctxs]
    yield L, capi, ctxs, pairs, addr, nbytes
    for c in ctxs:
        c.close()


def read_pair(L, capi, ctx, vec):
    v, e = C.c_double(), C.c_double()
    capi.check(L.abft_hip_read_pair(ctx.h, vec.device_ptr, C.byref(v), C.byref(e)))
    return v.value, e.value


def test_two_ranks_sum_in_rank_order_and_repeat(pair_of_contexts):
    L, capi, ctxs, pairs, addr, nbytes = pair_of_contexts
    for r, c in enumerate(ctxs):
        capi.check(L.abft_hip_peer_board_attach(c.h, addr[r], nbytes, r, 2, 20.0))
    rng = np.random.default_rng(5)
    for k in range(7):  # odd and even sequence numbers: both rows of the board
        mine = [np.array([rng.standard_normal() * 10.0 ** rng.integers(-8, 8), float(rng.integers(0, 5))])
                for _ in ctxs]
        for c, p, m in zip(ctxs, pairs, mine):
            c.upload(p, m)
        for c, p in zip(ctxs, pairs):  # enqueue-only: rank 0's kernel waits on the GPU for rank 1's
            capi.check(L.abft_hip_allreduce_pair_peers(c.h, p.device_ptr))
        got = [read_pair(L, capi, c, p) for c, p in zip(ctxs, pairs)]
        want = (0.0 + mine[0][0] + mine[1][0], mine[0][1] + mine[1][1])
        assert got[0] == got[1] == want, k
        assert not any(L.abft_hip_peer_board_failed(c.h) for c in ctxs)


def test_inside_a_replayed_graph(pair_of_contexts):
    L, capi, ctxs, pairs, addr, nbytes = pair_of_contexts
    for r, c in enumerate(ctxs):
        capi.check(L.abft_hip_peer_board_attach(c.h, addr[r], nbytes, r, 2, 20.0))
    graphs = []
    for c, p in zip(ctxs, pairs):
        g = C.c_void_p()
        capi.check(L.abft_hip_graph_begin(c.h))
        capi.check(L.abft_hip_allreduce_pair_peers(c.h, p.device_ptr))
        capi.check(L.abft_hip_graph_end(c.h, C.byref(g)))
        graphs.append(g)
    for c, p, m in zip(ctxs, pairs, ([1.0, 1.0], [2.5, 0.0])):
        c.upload(p, np.array(m))
    want = [1.0, 1.0], [2.5, 0.0]
    for k in range(4):  # in place: every replay sums the previous sums (the sequence number lives on the device)
        for g in graphs:
            capi.check(L.abft_hip_graph_launch(g))
        got = [read_pair(L, capi, c, p) for c, p in zip(ctxs, pairs)]
        s = (want[0][0] + want[1][0], want[0][1] + want[1][1])
        assert got[0] == got[1] == s, k
        want = [list(s), list(s)]
    for g in graphs:
        L.abft_hip_graph_destroy(g)


def test_a_missing_peer_is_a_bounded_wait_and_a_loud_result(pair_of_contexts):
    L, capi, ctxs, pairs, addr, nbytes = pair_of_contexts
    capi.check(L.abft_hip_peer_board_attach(ctxs[0].h, addr[0], nbytes, 0, 2, 0.5))  # rank 1 never shows up
    ctxs[0].upload(pairs[0], np.array([3.0, 0.0]))
    capi.check(L.abft_hip_allreduce_pair_peers(ctxs[0].h, pairs[0].device_ptr))
    v, _ = read_pair(L, capi, ctxs[0], pairs[0])
    assert math.isnan(v) and L.abft_hip_peer_board_failed(ctxs[0].h) == 1


def test_attach_refuses_bad_arguments(pair_of_contexts):
    L, capi, ctxs, pairs, addrs, nbytes = pair_of_contexts
    c, addr = ctxs[0], addrs[0]
    assert L.abft_hip_allreduce_pair_peers(c.h, pairs[0].device_ptr) != 0  # not attached
    assert L.abft_hip_peer_board_attach(c.h, addr, nbytes - 4096, 0, 2, 1.0) != 0
    assert L.abft_hip_peer_board_attach(c.h, addr + 8, nbytes, 0, 2, 1.0) != 0
    assert L.abft_hip_peer_board_attach(c.h, addr, nbytes, 2, 2, 1.0) != 0
    assert L.abft_hip_peer_board_attach(c.h, addr, nbytes, 0, 65, 1.0) != 0
    capi.check(L.abft_hip_peer_board_attach(c.h, addr, nbytes, 0, 1, 1.0))
    assert L.abft_hip_peer_board_attach(c.h, addr, nbytes, 0, 1, 1.0) != 0  # twice
    c.upload(pairs[0], np.array([7.25, 2.0]))
    capi.check(L.abft_hip_allreduce_pair_peers(c.h, pairs[0].device_ptr))  # one rank: the identity (0.0 + v)
    assert read_pair(L, capi, c, pairs[0]) == (7.25, 2.0)
    capi.check(L.abft_hip_peer_board_detach(c.h))
    capi.check(L.abft_hip_peer_board_detach(c.h))
