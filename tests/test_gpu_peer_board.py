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


# ---- window exchange (abft_hip_peer_exchange_*) ----

class Piece(C.Structure):
    _fields_ = [("peer", C.c_int), ("vector_offset", C.c_uint32), ("count", C.c_uint32), ("box_offset", C.c_uint64)]


def pieces(items):
    arr = (Piece * max(1, len(items)))()
    for a, (peer, voff, count, boff) in zip(arr, items):
        a.peer, a.vector_offset, a.count, a.box_offset = peer, voff, count, boff
    return arr


@pytest.fixture()
def two_ranks_with_outboxes():
    import abft_sparse_cg_amd as amd
    from abft_sparse_cg_amd import capi
    L = capi.load()
    box = 3 * 4096  # bytes per outbox
    nbytes = L.abft_hip_peer_exchange_bytes(2, box)
    fd = os.memfd_create("abft_xchg_test")
    os.ftruncate(fd, nbytes)
    maps = [mmap.mmap(fd, nbytes), mmap.mmap(fd, nbytes)]
    os.close(fd)
    addr = [C.addressof(C.c_char.from_buffer(m)) for m in maps]
    ctxs = [amd.HIPContext("none", "csr"), amd.HIPContext("none", "csr")]
    yield L, capi, ctxs, addr, nbytes, box
    for c in ctxs:
        c.close()


def test_windows_travel_both_ways_round_after_round(two_ranks_with_outboxes):
    L, capi, ctxs, addr, nbytes, box = two_ranks_with_outboxes
    slot = 1000
    # rank 0 reads [0, 37) and [500, 501) of rank 1's slot; rank 1 reads [990, 1000) of rank 0's
    out = [pieces([(1, 0 * slot + 990, 10, 0)]), pieces([(0, 1 * slot + 0, 37, 0), (0, 1 * slot + 500, 1, 512)])]
    inn = [pieces([(1, 1 * slot + 0, 37, 0), (1, 1 * slot + 500, 1, 512)]), pieces([(0, 0 * slot + 990, 10, 0)])]
    nout, nin = [1, 2], [2, 1]
    for r, c in enumerate(ctxs):
        capi.check(L.abft_hip_peer_exchange_attach(c.h, addr[r], nbytes, r, 2, box, out[r], nout[r], inn[r], nin[r], 20.0))
    full = [c.create_vector(2 * slot) for c in ctxs]
    rng = np.random.default_rng(11)
    for k in range(5):
        mine = [rng.standard_normal(slot) for _ in ctxs]
        before = []
        for r, (c, v) in enumerate(zip(ctxs, full)):
            h = np.full(2 * slot, -7.0 - k)
            h[r * slot:(r + 1) * slot] = mine[r]
            c.upload(v, h)
            before.append(h)
        for c, v in zip(ctxs, full):
            capi.check(L.abft_hip_peer_exchange(c.h, v.h))
        got = [c.download(v) for c, v in zip(ctxs, full)]
        want0, want1 = before[0].copy(), before[1].copy()
        want0[slot:slot + 37] = mine[1][:37]
        want0[slot + 500] = mine[1][500]
        want1[990:1000] = mine[0][990:]
        assert np.array_equal(got[0], want0) and np.array_equal(got[1], want1), k
        assert not any(L.abft_hip_peer_exchange_failed(c.h) for c in ctxs)


def test_windows_pushed_through_device_memory(two_ranks_with_outboxes):
    """Round 3: the same exchange through DEVICE memory -- a region per rank, every window pushed into the
    reader's region, every wait and read on the rank's own memory (between GPUs: xGMI stores; here the regions
    are plain device pointers of two contexts of one process, across processes they travel as IPC handles:
    test_gpu_cli.py).  Both ways and one way (the back-pressure word pushed to the sender), round after round,
    and a missing peer is a bounded wait with a loud result."""
    L, capi, ctxs, addr, nbytes, box = two_ranks_with_outboxes
    regions = (C.c_void_p * 2)()
    for r, c in enumerate(ctxs):
        p = C.c_void_p()
        capi.check(L.abft_hip_peer_exchange_device_alloc(c.h, 2, box, C.byref(p)))
        regions[r] = p.value
    slot = 1000
    out = [pieces([(1, 0 * slot + 990, 10, 0)]), pieces([(0, 1 * slot + 0, 37, 0), (0, 1 * slot + 500, 1, 512)])]
    inn = [pieces([(1, 1 * slot + 0, 37, 0), (1, 1 * slot + 500, 1, 512)]), pieces([(0, 0 * slot + 990, 10, 0)])]
    nout, nin = [1, 2], [2, 1]
    for r, c in enumerate(ctxs):
        capi.check(L.abft_hip_peer_exchange_attach_device(c.h, regions, r, 2, box, out[r], nout[r], inn[r], nin[r], 20.0))
    assert L.abft_hip_peer_exchange_attach_device(ctxs[0].h, regions, 0, 2, box, out[0], 1, inn[0], 2, 1.0) != 0  # twice
    full = [c.create_vector(2 * slot) for c in ctxs]
    rng = np.random.default_rng(11)
    for k in range(6):
        mine = [rng.standard_normal(slot) for _ in ctxs]
        before = []
        for r, (c, v) in enumerate(zip(ctxs, full)):
            h = np.full(2 * slot, -7.0 - k)
            h[r * slot:(r + 1) * slot] = mine[r]
            c.upload(v, h)
            before.append(h)
        for c, v in zip(ctxs, full):
            capi.check(L.abft_hip_peer_exchange(c.h, v.h))
        got = [c.download(v) for c, v in zip(ctxs, full)]
        want0, want1 = before[0].copy(), before[1].copy()
        want0[slot:slot + 37] = mine[1][:37]
        want0[slot + 500] = mine[1][500]
        want1[990:1000] = mine[0][990:]
        assert np.array_equal(got[0], want0) and np.array_equal(got[1], want1), k
        assert not any(L.abft_hip_peer_exchange_failed(c.h) for c in ctxs)
    for c in ctxs:
        capi.check(L.abft_hip_peer_exchange_detach(c.h))
    # one way: rank 0 only sends, rank 1 only receives (rank 0 asks rank 1's `done` word before reusing a box)
    n = 300
    capi.check(L.abft_hip_peer_exchange_attach_device(ctxs[0].h, regions, 0, 2, box, pieces([(1, 5, n, 0)]), 1, pieces([]), 0, 20.0))
    capi.check(L.abft_hip_peer_exchange_attach_device(ctxs[1].h, regions, 1, 2, box, pieces([]), 0, pieces([(0, 5, n, 0)]), 1, 20.0))
    small = [c.create_vector(1024) for c in ctxs]
    for k in range(6):
        src = rng.standard_normal(1024)
        ctxs[0].upload(small[0], src)
        ctxs[1].upload(small[1], np.zeros(1024))
        capi.check(L.abft_hip_peer_exchange(ctxs[0].h, small[0].h))
        capi.check(L.abft_hip_peer_exchange(ctxs[1].h, small[1].h))
        want = np.zeros(1024)
        want[5:5 + n] = src[5:5 + n]
        assert np.array_equal(ctxs[1].download(small[1]), want), k
    assert not any(L.abft_hip_peer_exchange_failed(c.h) for c in ctxs)
    for c in ctxs:
        capi.check(L.abft_hip_peer_exchange_detach(c.h))
    # a peer that never shows up
    capi.check(L.abft_hip_peer_exchange_attach_device(ctxs[0].h, regions, 0, 2, box, pieces([(1, 0, 8, 0)]), 1,
                                                      pieces([(1, 32, 8, 0)]), 1, 0.5))
    v = ctxs[0].create_vector(64)
    ctxs[0].upload(v, np.arange(64.0))
    capi.check(L.abft_hip_peer_exchange(ctxs[0].h, v.h))
    got = ctxs[0].download(v)
    assert math.isnan(got[32]) and L.abft_hip_peer_exchange_failed(ctxs[0].h) == 1
    capi.check(L.abft_hip_peer_exchange_detach(ctxs[0].h))
    for r, c in enumerate(ctxs):
        capi.check(L.abft_hip_peer_exchange_device_free(c.h, regions[r]))


def test_exchange_gives_up_loudly_and_refuses_bad_windows(two_ranks_with_outboxes):
    L, capi, ctxs, addr, nbytes, box = two_ranks_with_outboxes
    c = ctxs[0]
    v = c.create_vector(64)
    assert L.abft_hip_peer_exchange(c.h, v.h) != 0  # not attached
    one = pieces([(1, 0, 8, 0)])
    assert L.abft_hip_peer_exchange_attach(c.h, addr[0], nbytes, 0, 2, box, pieces([(0, 0, 8, 0)]), 1, one, 1, 1.0) != 0  # to itself
    assert L.abft_hip_peer_exchange_attach(c.h, addr[0], nbytes, 0, 2, box, pieces([(1, 0, 8, 4)]), 1, one, 1, 1.0) != 0  # unaligned
    assert L.abft_hip_peer_exchange_attach(c.h, addr[0], nbytes, 0, 2, box, pieces([(1, 0, box // 8, 0)]), 1, one, 1, 1.0) != 0  # no room for the check word
    assert L.abft_hip_peer_exchange_attach(c.h, addr[0], nbytes - 4096, 0, 2, box, one, 1, one, 1, 1.0) != 0
    capi.check(L.abft_hip_peer_exchange_attach(c.h, addr[0], nbytes, 0, 2, box, pieces([(1, 0, 8, 0)]), 1,
                                               pieces([(1, 32, 8, 0)]), 1, 0.5))
    small = c.create_vector(16)
    assert L.abft_hip_peer_exchange(c.h, small.h) != 0  # the windows reach past this vector
    c.upload(v, np.arange(64.0))
    capi.check(L.abft_hip_peer_exchange(c.h, v.h))  # rank 1 never shows up
    got = c.download(v)
    assert math.isnan(got[32]) and L.abft_hip_peer_exchange_failed(c.h) == 1
    capi.check(L.abft_hip_peer_exchange_detach(c.h))


def test_all_reduce_in_the_tail_of_the_reductions(pair_of_contexts):
    """abft_hip_peer_board_fuse: dot_dev / calc_xr_dev / the product of spmv_dot_dev deliver {sum, events}
    already summed over the ranks -- same bits as the reduction followed by the all-reduce kernel"""
    L, capi, ctxs, pairs, addr, nbytes = pair_of_contexts
    for r, c in enumerate(ctxs):
        capi.check(L.abft_hip_peer_board_attach(c.h, addr[r], nbytes, r, 2, 20.0))
    rng = np.random.default_rng(3)
    n = [70001, 1234]  # several blocks on one rank, one on the other
    vecs = []
    for c, m in zip(ctxs, n):
        a, b = c.create_vector(m), c.create_vector(m)
        c.upload(a, rng.standard_normal(m))
        c.upload(b, rng.standard_normal(m))
        vecs.append((a, b))

    def both(fused):
        for c in ctxs:
            capi.check(L.abft_hip_peer_board_fuse(c.h, 1 if fused else 0))
        for c, p, (a, b) in zip(ctxs, pairs, vecs):
            capi.check(L.abft_hip_dot_dev(c.h, a.h, b.h, p.device_ptr))
            if not fused:
                capi.check(L.abft_hip_allreduce_pair_peers(c.h, p.device_ptr))
        return [read_pair(L, capi, c, p) for c, p in zip(ctxs, pairs)]

    plain = both(False)
    fused = both(True)
    again = both(False)
    assert plain[0] == plain[1] and fused == plain and again == plain
    for c in ctxs:
        capi.check(L.abft_hip_peer_board_fuse(c.h, 0))
    assert L.abft_hip_peer_board_fuse(ctxs[0].h, 1) == 0 and L.abft_hip_peer_board_detach(ctxs[0].h) == 0
    assert L.abft_hip_peer_board_fuse(ctxs[0].h, 1) != 0  # not attached any more


def attach_device_boards(L, capi, ctxs, timeout=20.0, ranks=None):
    """the board in device memory, one copy per rank: inside one process the copies are plain pointers
    (across processes they travel as IPC handles: test_gpu_cli.py)"""
    size = len(ctxs)
    boards = (C.c_void_p * size)()
    for r, c in enumerate(ctxs):
        p = C.c_void_p()
        capi.check(L.abft_hip_peer_board_device_alloc(c.h, C.byref(p)))
        boards[r] = p.value
    for r, c in enumerate(ctxs):
        if ranks is None or r in ranks:
            capi.check(L.abft_hip_peer_board_attach_device(c.h, boards, r, size, timeout))
    return boards


def test_device_memory_board_sums_like_the_host_memory_board(pair_of_contexts):
    """Round 3: every rank keeps a copy of the board in its own device memory, pushes its slot into every
    copy and polls its own (between GPUs: stores over xGMI, no host memory).  Same sums, same bits on both
    ranks, both rows of the board, inside a replayed graph, in the tails of the reductions, and a missing
    peer is still a bounded wait with a loud result."""
    L, capi, ctxs, pairs, addr, nbytes = pair_of_contexts
    boards = attach_device_boards(L, capi, ctxs)
    assert L.abft_hip_peer_board_attach_device(ctxs[0].h, boards, 0, 2, 1.0) != 0  # twice
    rng = np.random.default_rng(5)
    for k in range(7):
        mine = [np.array([rng.standard_normal() * 10.0 ** rng.integers(-8, 8), float(rng.integers(0, 5))]) for _ in ctxs]
        for c, p, m in zip(ctxs, pairs, mine):
            c.upload(p, m)
        for c, p in zip(ctxs, pairs):
            capi.check(L.abft_hip_allreduce_pair_peers(c.h, p.device_ptr))
        got = [read_pair(L, capi, c, p) for c, p in zip(ctxs, pairs)]
        assert got[0] == got[1] == (0.0 + mine[0][0] + mine[1][0], mine[0][1] + mine[1][1]), k
        assert not any(L.abft_hip_peer_board_failed(c.h) for c in ctxs)
    # replayed
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
    for k in range(4):
        for g in graphs:
            capi.check(L.abft_hip_graph_launch(g))
        got = [read_pair(L, capi, c, p) for c, p in zip(ctxs, pairs)]
        s = (want[0][0] + want[1][0], want[0][1] + want[1][1])
        assert got[0] == got[1] == s, k
        want = [list(s), list(s)]
    for g in graphs:
        L.abft_hip_graph_destroy(g)
    # in the tails of the reductions
    vecs = []
    for c, m in zip(ctxs, (70001, 1234)):
        a, b = c.create_vector(m), c.create_vector(m)
        c.upload(a, rng.standard_normal(m))
        c.upload(b, rng.standard_normal(m))
        vecs.append((a, b))

    def both(fused):
        for c in ctxs:
            capi.check(L.abft_hip_peer_board_fuse(c.h, 1 if fused else 0))
        for c, p, (a, b) in zip(ctxs, pairs, vecs):
            capi.check(L.abft_hip_dot_dev(c.h, a.h, b.h, p.device_ptr))
            if not fused:
                capi.check(L.abft_hip_allreduce_pair_peers(c.h, p.device_ptr))
        return [read_pair(L, capi, c, p) for c, p in zip(ctxs, pairs)]

    plain, fused = both(False), both(True)
    assert plain[0] == plain[1] and fused == plain
    for c in ctxs:
        capi.check(L.abft_hip_peer_board_fuse(c.h, 0))
        capi.check(L.abft_hip_peer_board_detach(c.h))
    # a peer that never shows up
    capi.check(L.abft_hip_peer_board_attach_device(ctxs[0].h, boards, 0, 2, 0.5))
    ctxs[0].upload(pairs[0], np.array([3.0, 0.0]))
    # (the previous rounds' slots are still on the board with other sequence numbers: none fits a fresh counter)
    capi.check(L.abft_hip_allreduce_pair_peers(ctxs[0].h, pairs[0].device_ptr))
    v, _ = read_pair(L, capi, ctxs[0], pairs[0])
    assert math.isnan(v) and L.abft_hip_peer_board_failed(ctxs[0].h) == 1
    capi.check(L.abft_hip_peer_board_detach(ctxs[0].h))
    for r, c in enumerate(ctxs):
        capi.check(L.abft_hip_peer_board_device_free(c.h, boards[r]))


def test_one_way_windows_wait_for_the_reader_before_reusing_an_outbox(two_ranks_with_outboxes):
    """rank 0 sends, rank 1 only receives: nothing rank 0 waits for tells it that rank 1 is through
    with an outbox, so from the third exchange on it asks (the `done` word) before overwriting one"""
    L, capi, ctxs, addr, nbytes, box = two_ranks_with_outboxes
    n = 300
    capi.check(L.abft_hip_peer_exchange_attach(ctxs[0].h, addr[0], nbytes, 0, 2, box, pieces([(1, 5, n, 0)]), 1,
                                               pieces([]), 0, 20.0))
    capi.check(L.abft_hip_peer_exchange_attach(ctxs[1].h, addr[1], nbytes, 1, 2, box, pieces([]), 0,
                                               pieces([(0, 5, n, 0)]), 1, 20.0))
    full = [c.create_vector(1024) for c in ctxs]
    rng = np.random.default_rng(2)
    for k in range(6):
        src = rng.standard_normal(1024)
        ctxs[0].upload(full[0], src)
        ctxs[1].upload(full[1], np.zeros(1024))
        # the sender first, twice as often as not ahead of the reader
        capi.check(L.abft_hip_peer_exchange(ctxs[0].h, full[0].h))
        capi.check(L.abft_hip_peer_exchange(ctxs[1].h, full[1].h))
        got = ctxs[1].download(full[1])
        want = np.zeros(1024)
        want[5:5 + n] = src[5:5 + n]
        assert np.array_equal(got, want), k
        assert np.array_equal(ctxs[0].download(full[0]), src)
    assert not any(L.abft_hip_peer_exchange_failed(c.h) for c in ctxs)


EIGHT_RANKS = r'''
import ctypes as C, sys
import numpy as np
import abft_sparse_cg_amd as amd
from abft_sparse_cg_amd import capi
L = capi.load()
G = 8
ctxs = [amd.HIPContext("none", "csr") for _ in range(G)]
pairs = [c.create_vector(2) for c in ctxs]
boards = (C.c_void_p * G)()
for r, c in enumerate(ctxs):
    p = C.c_void_p()
    capi.check(L.abft_hip_peer_board_device_alloc(c.h, C.byref(p)))
    boards[r] = p.value
for r, c in enumerate(ctxs):
    capi.check(L.abft_hip_peer_board_attach_device(c.h, boards, r, G, 30.0))
def read(c, v):
    a, e = C.c_double(), C.c_double()
    capi.check(L.abft_hip_read_pair(c.h, v.device_ptr, C.byref(a), C.byref(e)))
    return a.value, e.value
rng = np.random.default_rng(8)
for k in range(6):  # both rows of the board, three times
    mine = [np.array([rng.standard_normal() * 10.0 ** rng.integers(-8, 8), float(rng.integers(0, 3))]) for _ in ctxs]
    for c, p, m in zip(ctxs, pairs, mine):
        c.upload(p, m)
    for c, p in zip(reversed(ctxs), reversed(pairs)):  # the last rank first: arrival order must not matter
        capi.check(L.abft_hip_allreduce_pair_peers(c.h, p.device_ptr))
    got = [read(c, p) for c, p in zip(ctxs, pairs)]
    s, e = 0.0, 0.0
    for m in mine:  # rank order
        s += m[0]; e += m[1]
    assert all(g == (s, e) for g in got), (k, got, (s, e))
# in the tails of the reductions, every rank a different length
vecs = []
for r, c in enumerate(ctxs):
    m = 1000 + 9001 * r
    a, b = c.create_vector(m), c.create_vector(m)
    c.upload(a, rng.standard_normal(m)); c.upload(b, rng.standard_normal(m))
    vecs.append((a, b))
def both(fused):
    for c in ctxs:
        capi.check(L.abft_hip_peer_board_fuse(c.h, 1 if fused else 0))
    for c, p, (a, b) in zip(ctxs, pairs, vecs):
        capi.check(L.abft_hip_dot_dev(c.h, a.h, b.h, p.device_ptr))
        if not fused:
            capi.check(L.abft_hip_allreduce_pair_peers(c.h, p.device_ptr))
    return [read(c, p) for c, p in zip(ctxs, pairs)]
plain, fused = both(False), both(True)
assert len(set(plain)) == 1 and fused == plain, (plain, fused)
assert not any(L.abft_hip_peer_board_failed(c.h) for c in ctxs)
# the window exchange with 8 ranks: a ring of halos, every rank reads 5 entries of each neighbour's slot
box = 4096
regions = (C.c_void_p * G)()
for r, c in enumerate(ctxs):
    p = C.c_void_p()
    capi.check(L.abft_hip_peer_exchange_device_alloc(c.h, G, box, C.byref(p)))
    regions[r] = p.value
class Piece(C.Structure):
    _fields_ = [("peer", C.c_int), ("vector_offset", C.c_uint32), ("count", C.c_uint32), ("box_offset", C.c_uint64)]
def pieces(items):
    arr = (Piece * max(1, len(items)))()
    for a, (peer, voff, count, boff) in zip(arr, items):
        a.peer, a.vector_offset, a.count, a.box_offset = peer, voff, count, boff
    return arr
slot = 100
for r, c in enumerate(ctxs):
    lo, hi = (r - 1) % G, (r + 1) % G
    # my outbox: the head of my slot for the rank below (box offset 0), the tail for the rank above (offset 256);
    # a rank lists its pairs in ascending peer order
    out = sorted([(lo, r * slot, 5, 0), (hi, r * slot + slot - 5, 5, 256)])
    # I read the tail of the slot below (it sits at offset 256 of ITS outbox) and the head of the slot above (offset 0)
    inn = sorted([(lo, lo * slot + slot - 5, 5, 256), (hi, hi * slot, 5, 0)])
    capi.check(L.abft_hip_peer_exchange_attach_device(c.h, regions, r, G, box, pieces(out), 2, pieces(inn), 2, 30.0))
full = [c.create_vector(G * slot) for c in ctxs]
for k in range(4):
    mine = [rng.standard_normal(slot) for _ in ctxs]
    for r, (c, v) in enumerate(zip(ctxs, full)):
        h = np.full(G * slot, -1.0 - k)
        h[r * slot:(r + 1) * slot] = mine[r]
        c.upload(v, h)
    for c, v in zip(reversed(ctxs), reversed(full)):
        capi.check(L.abft_hip_peer_exchange(c.h, v.h))
    for r, (c, v) in enumerate(zip(ctxs, full)):
        got = c.download(v)
        lo, hi = (r - 1) % G, (r + 1) % G
        want = np.full(G * slot, -1.0 - k)
        want[r * slot:(r + 1) * slot] = mine[r]
        want[lo * slot + slot - 5:lo * slot + slot] = mine[lo][-5:]
        want[hi * slot:hi * slot + 5] = mine[hi][:5]
        assert np.array_equal(got, want), (k, r)
assert not any(L.abft_hip_peer_exchange_failed(c.h) for c in ctxs)
for c in ctxs:
    c.close()
print("eight ranks ok")
'''


def test_board_and_windows_with_eight_ranks():
    """G = 8: the device-memory board (plain all-reduce, both rows; in the tails of the
    reductions) and the pushed window exchange (a ring of halos) with EIGHT contexts of one process standing in
    for eight ranks.  (Eight processes cannot share the one GPU of this pool's boxes: at most six may use a card
    at once; five do beside the test process, at configs[3]'s full size, in test_gpu_cli.py.)  In a child process: eight contexts whose
    kernels wait for each other need eight hardware queues (GPU_MAX_HW_QUEUES; the runtime's default is four
    per process, read when it initialises)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", PYTHONPATH=root)
    p = subprocess.run([sys.executable, "-c", EIGHT_RANKS], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0 and "eight ranks ok" in p.stdout, p.stdout[-800:] + p.stderr[-3000:]
