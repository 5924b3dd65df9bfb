"""Row-block sharded CG over torch.distributed (one process per GPU; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" on CPU for the tests).

The reference is single-process (no counterpart); the decomposition
is the one the north star names:

  * rank g owns a contiguous row block [b_g, b_{g+1}) (cut by nnz, not rows) of
    the CSR arrays and the matching slices of x, r, p, w, b;
  * SpMV needs the whole search vector: the local slices live in equal-sized
    slots of one gathered buffer (slot g at [g*S, g*S + rows_g)); column indices
    are re-based to that padded layout when the shard is created, so the kernel
    gathers from the buffer directly.  Per iteration the slots are filled either
    by one all_gather_into_tensor, or -- when every rank only reads a window of
    its peers' slots (banded matrices: the halo) -- by point-to-point copies of
    just those windows;
  * dot and calc_xr produce {partial sum, queued-event count} on the device;
    one all-reduce(sum) of those two doubles gives every rank the scalar and
    tells it whether any rank has an ECC event to report;
  * calc_p / ECC checks / write-backs are purely local.  Event indices are
    global (index_base + local element index); events are gathered to rank 0,
    which prints them in index order; a fatal one ends every rank with status 1.

The compute engine is a parameter: the product passes HipEngine (the C ABI);
tests pass a CPU stand-in to exercise the partition/collective logic under gloo.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .context import fdiv


class _DevMem:
    """Zero-copy view of library-owned device memory for torch (CUDA array interface)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class HipEngine:
    """The product engine: every operation is a libabft_hip.so call on this
    rank's GPU; tensors alias the library's device buffers for the collectives."""

    def __init__(self, mode, fmt="csr", device=0):
        from .context import HIPContext
        self.device = device
        torch.cuda.set_device(device)
        self.pending = []
        self.ctx = HIPContext(mode, fmt, device=device, on_event=lambda ev, fatal: self.pending.extend(ev))
        # One stream for everything: make a torch stream this thread's current
        # stream (torch.distributed enqueues collectives relative to it) and run the
        # library's kernels on the same stream, so kernels and collectives are
        # ordered by the stream itself.  (The legacy default stream, handle 0, is
        # not usable for this: the library reads a null handle as "own stream".)
        self.tstream = torch.cuda.Stream(device=device)
        torch.cuda.set_stream(self.tstream)
        assert self.tstream.cuda_stream != 0
        capi.check(self.ctx.L.abft_hip_set_stream(self.ctx.h, self.tstream.cuda_stream))
        self.L = self.ctx.L

    def create_matrix(self, cols, rows, vals, n_out, n_in, index_base):
        return self.ctx.create_matrix(cols, rows, vals, n_out, len(vals), n_in=n_in, index_base=index_base)

    def create_vector(self, n):
        return self.ctx.create_vector(n)

    def view(self, parent, off, n):
        return self.ctx.view_vector(parent, off, n)

    def tensor(self, vec):
        return torch.as_tensor(_DevMem(vec.device_ptr, max(vec.N, 1)), device="cuda:%d" % self.device)[:vec.N]

    def upload(self, vec, arr):
        self.ctx.upload(vec, arr)

    def download(self, vec):
        return self.ctx.download(vec)

    def copy(self, dst, src):
        self.ctx.copy_vector(dst, src)

    def spmv(self, A, x, y, part=capi.PART_ALL):
        self.ctx.spmv(A, x, y, part)

    def set_interior(self, A, row_lo, row_hi):
        self.ctx.set_interior(A, row_lo, row_hi)

    def dot_partial(self, a, b, out):
        capi.check(self.L.abft_hip_dot_dev(self.ctx.h, a.h, b.h, out.device_ptr))

    def calc_xr_partial(self, x, r, p, w, alpha, out):
        capi.check(self.L.abft_hip_calc_xr_dev(self.ctx.h, x.h, r.h, p.h, w.h, alpha, out.device_ptr))

    def calc_p(self, p, r, beta):
        self.ctx.calc_p(p, r, beta)

    # device-scalar forms (include/abft_hip.h): nothing comes back to the host
    def spmv_dot(self, A, x, x_off, y, out, part=capi.PART_ALL):
        capi.check(self.L.abft_hip_spmv_dot_part_dev(self.ctx.h, A.h, x.h, y.h, x_off, out.device_ptr, part))

    def calc_xr_ratio(self, x, r, p, w, num, den, out):
        capi.check(self.L.abft_hip_calc_xr_ratio_dev(self.ctx.h, x.h, r.h, p.h, w.h, num.device_ptr, den.device_ptr,
                                                     out.device_ptr))

    def calc_p_ratio(self, p, r, num, den):
        capi.check(self.L.abft_hip_calc_p_ratio_dev(self.ctx.h, p.h, r.h, num.device_ptr, den.device_ptr))

    def read_pair(self, vec):
        """{sum, events} at a 2-element device vector -> host, through the pinned slot the
        host polls (a few microseconds; a copy + stream synchronise costs tens)"""
        v, ev = C.c_double(), C.c_double()
        capi.check(self.L.abft_hip_read_pair(self.ctx.h, vec.device_ptr, C.byref(v), C.byref(ev)))
        return v.value, int(ev.value)

    def inject(self, A, index, bits):
        self.ctx.inject_at(A, index, bits)

    def drain(self):
        ev, _ = self.ctx.drain_events()
        ev = self.pending + ev
        self.pending = []
        return ev

    def synchronize(self):
        self.ctx.synchronize()

    def close(self):
        torch.cuda.current_stream().synchronize()
        self.ctx.close()
        torch.cuda.set_stream(torch.cuda.default_stream(self.device))


def pad_columns(cols, bounds, slot):
    """global column -> index in the slot-padded gathered vector"""
    b = np.asarray(bounds, dtype=np.int64)
    owner = np.searchsorted(b, cols, side="right") - 1
    return (owner * slot + (cols.astype(np.int64) - b[owner])).astype(np.uint32), owner


class ShardedCG:
    """One rank's share of a row-partitioned CG solve."""

    def __init__(self, engine, cols, rows, vals, bounds, nnz_before, mode, group=None, fmt_id=capi.FMT_CSR,
                 staged=False):
        """cols/rows/vals: this rank's rows (global indices, sorted by (row,col));
        bounds: the G+1 row-block boundaries; nnz_before: elements in lower ranks
        (the global index of this shard's first element).  staged=True runs every
        collective on host copies of the device buffers: for backends that cannot
        move device memory (gloo), e.g. several ranks sharing one GPU in tests."""
        self.e, self.group, self.mode, self.fmt_id = engine, group, mode, fmt_id
        self.staged = staged
        # measurement aid: issue the collectives even at world size 1 (their host-side cost is the same)
        self.force_coll = os.environ.get("ABFT_FORCE_COLLECTIVES") == "1"
        self._windows_by_alltoall = (dist.is_initialized() and not staged and dist.get_backend(group) == "nccl"
                                     and os.environ.get("ABFT_CG_WINDOWS", "alltoall") != "p2p")
        self._graph = None  # hipGraphs of the iteration (run_fixed); False once capture has failed
        self._warm = False
        self._replayed = False
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        assert len(bounds) == self.G + 1
        self.bounds = [int(b) for b in bounds]
        self.N = self.bounds[-1]
        self.r0, self.r1 = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.n_loc = self.r1 - self.r0
        self.slot = max(self.bounds[g + 1] - self.bounds[g] for g in range(self.G))
        self.n_pad = self.slot * self.G
        if mode not in ("none", "constraints") and self.n_pad > (1 << 24):
            raise ValueError("padded vector length %d exceeds the 24-bit column field of the ECC modes" % self.n_pad)
        cols = np.asarray(cols)
        if self.G == 1:
            pcols = cols.astype(np.uint32)
            owner = np.zeros(len(cols), dtype=np.int64)
        else:
            pcols, owner = pad_columns(cols, self.bounds, self.slot)
        lrows = (np.asarray(rows).astype(np.int64) - self.r0).astype(np.uint32)
        self.A = engine.create_matrix(pcols, lrows, vals, self.n_loc, self.n_pad, nnz_before)
        # Rows that read only this rank's own slot can be multiplied while the exchange
        # is in flight: take the longest run of consecutive such rows (for a banded
        # matrix: everything but the first and last few rows of the shard).  Row sums
        # are never split, so the result stays bit-identical.
        self.interior = None
        if self.G > 1 and self.n_loc:
            edge = np.concatenate(([-1], np.unique(lrows[owner != self.rank]).astype(np.int64), [self.n_loc]))
            k = int(np.argmax(np.diff(edge)))
            lo, hi = int(edge[k]) + 1, int(edge[k + 1])
            if hi - lo >= self.n_loc // 4:
                self.interior = (lo, hi)
                engine.set_interior(self.A, lo, hi)
        # which window of each peer's slot this rank reads
        need = []
        for g in range(self.G):
            m = owner == g
            if g == self.rank or not m.any():
                need.append((0, 0))
            else:
                lo = int(pcols[m].min()) - g * self.slot
                hi = int(pcols[m].max()) + 1 - g * self.slot
                need.append((lo, hi))
        self.need = need
        if self.G > 1:
            all_need = [None] * self.G
            dist.all_gather_object(all_need, need, group=group)
            self.all_need = all_need  # all_need[src][dst] = window of dst's slot that src reads
            total = sum(hi - lo for nd in all_need for lo, hi in nd)
            # windows are worth it when they move well under half of what the all-gather moves
            self.use_windows = total * 2 < self.N * (self.G - 1)
        else:
            self.use_windows = False
        # vectors: p lives inside the gathered buffer
        self.p_full = engine.create_vector(self.n_pad)
        self.p = engine.view(self.p_full, self.rank * self.slot, self.n_loc)
        self.x_full = None
        self.b, self.x, self.r, self.w = (engine.create_vector(self.n_loc) for _ in range(4))
        self.scal = engine.create_vector(2)
        self.t_full = engine.tensor(self.p_full)
        self.t_scal = engine.tensor(self.scal)
        engine.upload(self.p_full, np.zeros(self.n_pad))
        self.events = []
        self._p2p_cache = {}
        # interior rows beside the exchange only when the exchange is long enough to be worth
        # the asynchronous hand-off (a halo of a few KB is not: see exchange())
        incoming = 8 * (sum(hi - lo for lo, hi in self.need) if self.use_windows else (self.G - 1) * self.slot)
        self.overlap = bool(self.interior) and incoming >= int(os.environ.get("ABFT_CG_OVERLAP_BYTES", str(2 << 20)))
        if self.use_windows and self._windows_by_alltoall and not self._alltoall_selftest():
            self._windows_by_alltoall = False  # point-to-point copies instead

    def _alltoall_selftest(self):
        """One trial window exchange through all_to_all on a scratch vector (slot g filled with
        g + 1): every window received must carry its sender's number, on every rank.  A wrong
        answer or an exception on any rank sends all ranks to batch_isend_irecv."""
        ok = 1.0
        try:
            t = torch.zeros(self.n_pad, dtype=torch.float64, device=self.t_full.device)
            S, me = self.slot, self.rank
            t[me * S:me * S + self.n_loc] = float(me + 1)
            for req in self._window_exchange(t, None):
                req.wait()
            for g in range(self.G):
                lo, hi = self.need[g] if g != me else (0, 0)
                if hi > lo and not bool((t[g * S + lo:g * S + hi] == float(g + 1)).all()):
                    ok = 0.0
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write("abft: all_to_all window exchange failed its self-test (%s)\n" % exc)
            ok = 0.0
        flag = torch.tensor([ok], dtype=torch.float64, device=self.t_full.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        good = bool(flag.item() > 0.5)
        if not good and self.rank == 0:
            sys.stderr.write("abft: halo windows by batch_isend_irecv (all_to_all self-test failed)\n")
        return good

    # ---- collectives -------------------------------------------------------
    def exchange(self, full_vec_tensor):
        """Fill the peers' slots of a gathered buffer whose own slot is current -- the blocking
        (stream-ordered) forms of the collectives: the backend may issue them on the calling
        stream itself.  Their asynchronous forms (exchange_begin / exchange_finish) hand the
        work to the backend's stream and back, which costs ~25 us per exchange on this stack
        (measured: 91 -> 67 us per iteration on the 1/8 shard of config 2) and only pays when
        something worth more than that runs in between (self.overlap)."""
        if self.G == 1 and not self.force_coll:
            return
        if self.staged:
            self.exchange_finish(("staged", full_vec_tensor))
            return
        S, me = self.slot, self.rank
        if not self.use_windows:
            dist.all_gather_into_tensor(full_vec_tensor, full_vec_tensor[me * S:(me + 1) * S], group=self.group)
            return
        for req in self._window_exchange(full_vec_tensor, full_vec_tensor.data_ptr(), blocking=True):
            req.wait()

    def exchange_begin(self, full_vec_tensor):
        """Start the exchange and return a handle for exchange_finish.  Work that
        reads only this rank's own slot may be enqueued in between: the collective
        runs on the backend's stream beside it.  (Host-staged collectives are
        synchronous: there the whole exchange happens in exchange_finish, i.e.
        strictly after whatever was enqueued in between.)"""
        if self.G == 1 and not self.force_coll:
            return None
        if self.staged:
            return ("staged", full_vec_tensor)
        S, me = self.slot, self.rank
        if not self.use_windows:
            return [dist.all_gather_into_tensor(full_vec_tensor, full_vec_tensor[me * S:(me + 1) * S],
                                                group=self.group, async_op=True)]
        return self._window_exchange(full_vec_tensor, full_vec_tensor.data_ptr())

    def exchange_finish(self, handle):
        if handle is None:
            return
        if isinstance(handle, tuple):  # host-staged
            dev_tensor = handle[1]
            host = dev_tensor.cpu()
            S, me = self.slot, self.rank
            if not self.use_windows:
                dist.all_gather_into_tensor(host, host[me * S:(me + 1) * S].clone(), group=self.group)
            else:
                for req in self._window_exchange(host, None):
                    req.wait()
            dev_tensor.copy_(host)
            return
        for req in handle:
            req.wait()

    def _window_lists(self, t, key):
        """Per rank g: what this rank receives from g (a window of g's slot of `t`) and what
        it sends to g (the window of its own slot that g reads); empty tensors where nothing
        moves.  The windows are fixed for the life of the solver: built once per buffer."""
        lists = self._p2p_cache.get(key) if key is not None else None
        if lists is None:
            S, me = self.slot, self.rank
            recv, send = [], []
            for g in range(self.G):
                lo, hi = self.need[g] if g != me else (0, 0)          # what I read of rank g's slot
                recv.append(t[g * S + lo:g * S + hi])
                lo, hi = self.all_need[g][me] if g != me else (0, 0)  # what rank g reads of my slot
                send.append(t[me * S + lo:me * S + hi])
            lists = (recv, send)
            if key is not None:
                self._p2p_cache[key] = lists
        return lists

    def _window_exchange(self, t, key, blocking=False):
        """Start the window copies; returns the work objects to wait for.  On the nccl backend
        one all_to_all over the two lists (a single grouped send/receive: ~15 us to enqueue,
        against ~40 us and more for batch_isend_irecv); elsewhere (gloo has no all_to_all), or
        with ABFT_CG_WINDOWS=p2p, the same lists as point-to-point operations."""
        recv, send = self._window_lists(t, key)
        if not any(x.numel() for x in recv) and not any(x.numel() for x in send):
            return []
        if self._windows_by_alltoall:
            if blocking:
                dist.all_to_all(recv, send, group=self.group)
                return []
            return [dist.all_to_all(recv, send, group=self.group, async_op=True)]
        ops = []
        for g in range(self.G):
            if send[g].numel():
                ops.append(dist.P2POp(dist.isend, send[g], self._peer(g), self.group))
            if recv[g].numel():
                ops.append(dist.P2POp(dist.irecv, recv[g], self._peer(g), self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def _peer(self, g):
        return dist.get_global_rank(self.group, g) if self.group is not None else g

    def _allreduce_scalar(self):
        """-> (sum over ranks, total queued events); synchronises."""
        t = self.t_scal.cpu() if self.staged else self.t_scal
        if self.G > 1 or self.force_coll:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        if not self.staged and hasattr(self.e, "read_pair"):
            return self.e.read_pair(self.scal)  # ordered behind the collective on the shared stream
        v = t.tolist()
        return v[0], int(v[1])

    def _collect_events(self):
        """Gather every rank's queued events; rank 0 prints them in index order
        with the reference's text.  Returns True if one is fatal."""
        mine = self.e.drain()
        if self.G > 1:
            everyone = [None] * self.G
            dist.all_gather_object(everyone, mine, group=self.group)
            mine = [ev for lst in everyone for ev in lst]
        mine.sort(key=lambda ev: (ev[1], ev[0]))
        out, fatal = [], False
        for ev in mine:
            out.append(ev)
            if capi.is_fatal(ev[0]):
                fatal = True
                break
        self.events.extend(out)
        if self.rank == 0:
            for k, i, b in out:
                sys.stdout.write(capi.format_event(k, i, b, self.fmt_id))
            sys.stdout.flush()
        return fatal

    # ---- the solver --------------------------------------------------------
    def set_rhs(self, b_local, x_local=None):
        self.e.upload(self.b, b_local)
        self.e.upload(self.x, np.zeros(self.n_loc) if x_local is None else x_local)

    def dot(self, a, b):
        self.e.dot_partial(a, b, self.scal)
        v, nev = self._allreduce_scalar()
        if nev and self._collect_events():
            raise SystemExit(1)
        return v

    def start(self):
        """cg.cpp:87-91"""
        self.e.copy(self.r, self.b)
        self.e.copy(self.p, self.r)
        self.rr = self.dot(self.r, self.r)
        return self.rr

    def step(self):
        """One CG iteration, cg.cpp:97-114, with the exchange in front of the SpMV."""
        if self.overlap:
            h = self.exchange_begin(self.t_full)
            self.e.spmv(self.A, self.p_full, self.w, capi.PART_INTERIOR)  # rows that need no peer data
            self.exchange_finish(h)
            self.e.spmv(self.A, self.p_full, self.w, capi.PART_BOUNDARY)
        else:
            self.exchange(self.t_full)
            self.e.spmv(self.A, self.p_full, self.w)
        pw = self.dot(self.p, self.w)
        alpha = fdiv(self.rr, pw)
        self.e.calc_xr_partial(self.x, self.r, self.p, self.w, alpha, self.scal)
        rr_new, nev = self._allreduce_scalar()
        if nev and self._collect_events():
            raise SystemExit(1)
        beta = fdiv(rr_new, self.rr)
        self.e.calc_p(self.p, self.r, beta)
        self.rr = rr_new
        return rr_new

    def run_fixed(self, iters, graph=None):
        """`iters` CG iterations with no convergence test (-c 0), alpha and beta
        resident on the device: each iteration enqueues exchange, spmv+dot,
        all-reduce, calc_xr, all-reduce, calc_p and reads nothing back, so the host
        never waits inside the loop (at 8 GPUs the local kernels take ~40 us and a
        host round trip per scalar would dominate).  Same kernels, same arithmetic
        (alpha = rr/pw and beta = rr_new/rr formed in fp64 on the device) as step().

        graph (default: on at world size 1, off across ranks; ABFT_CG_GRAPH=1 / 0 decides
        otherwise; never with host-staged collectives):
        an iteration -- kernels, the two all-reduces and, in all-gather mode, the
        exchange -- is captured once into a hipGraph on the shared stream (two graphs:
        the rr / rr_new scalars swap roles every iteration) and replayed, so the host
        issues one launch per iteration instead of ~8 calls.  Point-to-point window
        copies stay outside the graph and are issued eagerly between replays (the
        interior rows' SpMV is replayed between their start and their completion).  The
        first two iterations of the first call run eagerly (communicators and peer
        connections are set up by their first use, which must not happen under
        capture); if capture is refused the loop carries on eagerly.

        ECC events are collected once, at the end.  Returns the final rr."""
        if not hasattr(self, "_pipe"):
            vecs = [self.e.create_vector(2) for _ in range(3)]
            self._pipe = [(v, self.e.tensor(v)) for v in vecs]
        if graph is None:
            # Default: replay at world size 1, where this stack is verified end to end; eager
            # enqueue across ranks unless ABFT_CG_GRAPH=1 asks for replay there too.  (RCCL
            # collectives inside a captured graph could only be exercised with one rank on the
            # development box, and a captured point-to-point copy crashed this torch/RCCL pair;
            # at 8 GPUs the eager enqueue costs about what the iteration takes on the GPU, so
            # little is lost by not risking it unasked.)
            env = os.environ.get("ABFT_CG_GRAPH")
            graph = (self.G == 1) if env is None else env != "0"
        # (HIP-event brackets around kernels cannot be captured: profiling turns the graph off)
        graph = graph and not self.staged and hasattr(self.e, "tstream") and not self.e.ctx.prof_mask
        s0, s1, pw = self._pipe
        self.e.copy(self.r, self.b)
        self.e.copy(self.p, self.r)
        self.e.dot_partial(self.r, self.r, s0[0])
        self._allreduce_async(s0[1])
        left, k = iters, 0
        if graph and not self._warm and left >= 2:
            self._iteration(s0, s1, pw)
            self._iteration(s1, s0, pw)
            self._warm = True
            left -= 2
        if graph and self._warm and left and self._graph is None:
            self._capture(s0, s1, pw)
        if graph and self._graph:
            while left:
                self._replay(k)
                k += 1
                left -= 1
        pair = (s0, s1)
        for _ in range(left):
            self._iteration(pair[k & 1], pair[1 - (k & 1)], pw)
            k += 1
        self._warm = True
        cur = pair[k & 1]
        v = (cur[1].cpu() if self.staged else cur[1]).tolist()  # the only synchronisation
        self.rr = v[0]
        if int(v[1]) and self._collect_events():
            raise SystemExit(1)
        return self.rr

    def _iteration(self, cur, nxt, pw):
        """enqueue one iteration: rr in cur -> rr_new in nxt"""
        if self.overlap:
            h = self.exchange_begin(self.t_full)
            self._interior_part(pw)
            self.exchange_finish(h)
        else:
            self.exchange(self.t_full)
        self._rest(cur, nxt, pw)

    def _interior_part(self, pw):
        # rows that read nothing from the peers: beside the exchange, not behind it
        self.e.spmv_dot(self.A, self.p_full, self.rank * self.slot, self.w, pw[0], capi.PART_INTERIOR)

    def _rest(self, cur, nxt, pw):
        self.e.spmv_dot(self.A, self.p_full, self.rank * self.slot, self.w, pw[0],
                        capi.PART_BOUNDARY if self.overlap else capi.PART_ALL)
        self._allreduce_async(pw[1])
        self.e.calc_xr_ratio(self.x, self.r, self.p, self.w, cur[0], pw[0], nxt[0])
        self._allreduce_async(nxt[1])
        self.e.calc_p_ratio(self.p, self.r, nxt[0], cur[0])

    def _capture(self, s0, s1, pw):
        """all-gather mode: one graph per parity holding the whole iteration;
        window mode: the point-to-point copies stay eager, so the iteration is cut
        at them into an interior graph and, per parity, a graph of the rest"""
        def record(fn, *args):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.e.tstream, capture_error_mode="thread_local"):
                fn(*args)
            return g
        try:
            if self.use_windows:
                gi = record(self._interior_part, pw) if self.overlap else None
                self._graph = (gi, [record(self._rest, s0, s1, pw), record(self._rest, s1, s0, pw)])
            else:
                self._graph = (None, [record(self._iteration, s0, s1, pw), record(self._iteration, s1, s0, pw)])
        except Exception as exc:  # capture refused: keep going without it
            sys.stderr.write("abft: hipGraph capture of the CG iteration failed (%s); running eagerly\n" % exc)
            self._graph = False
            torch.cuda.set_stream(self.e.tstream)

    def _replay(self, k):
        gi, rest = self._graph
        if self.use_windows:
            if self.overlap:
                h = self.exchange_begin(self.t_full)
                gi.replay()
                self.exchange_finish(h)
            else:
                self.exchange(self.t_full)
        rest[k & 1].replay()
        if not self._replayed:
            # first replay of a fresh capture: wait for it once, under a watchdog, so a
            # collective that cannot run from a graph on this stack ends the job with a
            # message within minutes instead of hanging it (ABFT_CG_GRAPH=0 avoids graphs)
            import threading
            def stuck():
                sys.stderr.write("abft: the first hipGraph replay of the CG iteration did not finish in 180 s; "
                                 "rerun with ABFT_CG_GRAPH=0\n")
                sys.stderr.flush()
                os._exit(70)
            t = threading.Timer(180.0, stuck)
            t.daemon = True
            t.start()
            self.e.tstream.synchronize()
            t.cancel()
            self._replayed = True

    def _allreduce_async(self, t):
        if self.G == 1 and not self.force_coll:
            return
        if self.staged:
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def solve(self, max_itrs=1000, conv_threshold=1e-3, on_iteration=None):
        rr = self.start()
        itr = 0
        while itr < max_itrs and rr > conv_threshold:
            rr = self.step()
            if on_iteration is not None:
                on_iteration(itr, rr)
            itr += 1
        return itr, rr

    def residual_check(self):
        """cg.cpp:127-144: r = A x, then total / max error against b (global)."""
        if self.x_full is None:
            self.x_full = self.e.create_vector(self.n_pad)
            self.e.upload(self.x_full, np.zeros(self.n_pad))
            self.t_xfull = self.e.tensor(self.x_full)
        xs = self.e.view(self.x_full, self.rank * self.slot, self.n_loc)
        self.e.copy(xs, self.x)
        self.exchange(self.t_xfull)
        self.e.spmv(self.A, self.x_full, self.r)
        ax = self.e.download(self.r)
        if self._collect_events():  # collective: every rank drains, once per solve
            raise SystemExit(1)
        err = np.abs(self.e.download(self.b) - ax)
        t = torch.tensor([float((err * err).sum()), float(err.max() if len(err) else 0.0)], dtype=torch.float64)
        if self.G > 1:
            dev = "cpu" if self.staged else self.t_scal.device
            s = t[:1].to(dev)
            m = t[1:].to(dev)
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
            t = torch.cat([s.cpu(), m.cpu()])
        return float(t[0]) ** 0.5, float(t[1])

    def gather_x(self):
        """The full solution on every rank (for checks)."""
        x = self.e.download(self.x)
        if self.G == 1:
            return x
        parts = [None] * self.G
        dist.all_gather_object(parts, x, group=self.group)
        return np.concatenate(parts)
