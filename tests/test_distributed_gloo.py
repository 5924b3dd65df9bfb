"""The N>1 path on CPU: world_size-2 (and 3) torch.distributed runs over gloo.

The partition / padded-slot layout / exchange / all-reduce / event plumbing of
abft_sparse_cg_amd.distributed.ShardedCG is the product code under test; the
per-rank compute engine is a test-only stand-in that calls the CPU oracle (the
HIP kernels need a GPU; tests/test_gpu_distributed.py runs the real engine).
Asserted: identical iteration count to the single-process oracle, rr history and
solution within 1e-10 relative, ECC events reported once with global indices."""
import os
import socket
import sys

import multiprocessing as mp

import numpy as np
import pytest

# torch is imported only inside the spawned ranks: keeping it out of the pytest
# process keeps that process small, which the fork-based reference tests rely on

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from _oracle import CSR, Oracle, OracleMatrix, f64p, _p, laplace5, random_spd, rhs  # noqa: E402


class _Vec:
    def __init__(self, arr):
        self.a = arr
        self.N = len(arr)


class OracleEngine:
    """Test-only engine: same interface as distributed.HipEngine, CPU oracle inside."""

    def __init__(self, mode):
        self.mode = mode
        self.L = Oracle.lib()
        self.mats = []

    def create_matrix(self, cols, rows, vals, n_out, n_in, index_base):
        m = OracleMatrix(CSR, self.mode, cols, rows, vals, n_out, n_in=n_in, index_base=index_base)
        self.mats.append(m)
        return m

    def create_vector(self, n):
        return _Vec(np.full(n, np.nan))

    def view(self, parent, off, n):
        return _Vec(parent.a[off:off + n])

    def tensor(self, vec):
        import torch
        return torch.from_numpy(vec.a)

    def upload(self, vec, arr):
        vec.a[:] = arr

    def download(self, vec):
        return vec.a.copy()

    def copy(self, dst, src):
        dst.a[:] = src.a[:dst.N]

    def set_interior(self, A, lo, hi):
        A.interior = (lo, hi)

    def spmv(self, A, x, y, part=0):
        """part 1 (interior) runs BEFORE the peers' data is known to have arrived, part
        2 (boundary) after: the interior rows computed early must equal, bit for bit,
        what a full SpMV gives once everything is in place -- i.e. the rows the
        solver declared interior really read nothing from a peer."""
        lo, hi = getattr(A, "interior", (0, 0))
        if part == 1 and hi <= lo:
            return
        tmp = np.full_like(y.a, np.nan)
        self.L.ora_spmv(A.h, _p(x.a, f64p), _p(tmp, f64p), 1)
        self._peek()
        fatal = any(k in (1, 4) or k >= 5 for k, _, _ in self._held)  # the SpMV stopped early: y is not meaningful
        if part == 1:
            y.a[lo:hi] = tmp[lo:hi]
            A.early = True
            return
        if part == 2 and hi > lo:
            assert getattr(A, "early", False), "boundary part without the interior part before it"
            assert fatal or np.array_equal(tmp[lo:hi].view(np.uint64), y.a[lo:hi].view(np.uint64)), \
                "an interior row changed when the peers' data arrived"
            A.early = False
        y.a[:] = tmp

    def dot_partial(self, a, b, out):
        out.a[0] = self.L.ora_dot(_p(a.a, f64p), _p(b.a, f64p), a.N)
        out.a[1] = self._peek()

    def calc_xr_partial(self, x, r, p, w, alpha, out):
        out.a[0] = self.L.ora_calc_xr(_p(x.a, f64p), _p(r.a, f64p), _p(p.a, f64p), _p(w.a, f64p), alpha, x.N)
        out.a[1] = self._peek()

    def calc_p(self, p, r, beta):
        self.L.ora_calc_p(_p(p.a, f64p), _p(r.a, f64p), beta, p.N)

    def spmv_dot(self, A, x, x_off, y, out, part=0):
        self.spmv(A, x, y, part)
        if part == 1:
            return
        xs = np.ascontiguousarray(x.a[x_off:x_off + y.N])
        out.a[0] = self.L.ora_dot(_p(xs, f64p), _p(y.a, f64p), y.N)
        out.a[1] = self._peek()

    def calc_xr_ratio(self, x, r, p, w, num, den, out):
        self.calc_xr_partial(x, r, p, w, num.a[0] / den.a[0], out)

    def calc_p_ratio(self, p, r, num, den):
        self.calc_p(p, r, num.a[0] / den.a[0])

    def inject(self, A, index, bits):
        A.inject(index, bits)

    def _peek(self):
        # queued-event count without draining (the oracle keeps it in the handle)
        self._held = getattr(self, "_held", [])
        for m in self.mats:
            ev, _ = m.events()
            self._held += ev
        return float(len(self._held))

    def drain(self):
        self._peek()
        ev, self._held = self._held, []
        return ev

    def synchronize(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # interior rows beside the exchange at any exchange size (the product only does so for long
    # exchanges), unless the case says otherwise: that is the path with the ordering property to check
    os.environ["ABFT_CG_OVERLAP_BYTES"] = str(case[8]) if len(case) > 8 else "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from abft_sparse_cg_amd.distributed import ShardedCG
        cols, rows, vals, n, bounds, mode, flip = case[:7]
        fixed = case[7] if len(case) > 7 else 0
        r0, r1 = bounds[rank], bounds[rank + 1]
        m = (rows >= r0) & (rows < r1)
        nnz_before = int(np.argmax(m)) if m.any() else 0
        eng = OracleEngine(mode)
        cg = ShardedCG(eng, cols[m], rows[m], vals[m], bounds, nnz_before, mode)
        if cg.use_windows:
            # the trial exchange the nccl path runs before trusting all_to_all (here over the
            # point-to-point form of the same window lists): every window carries its sender's number
            assert cg._alltoall_selftest()
        b = rhs(n, 1)
        cg.set_rhs(b[r0:r1])
        if flip is not None:
            gi, bits = flip
            if nnz_before <= gi < nnz_before + int(m.sum()):
                eng.inject(cg.A, gi - nnz_before, bits)
        hist = []
        code = 0
        if isinstance(mode, str) and mode.startswith("fixed:"):
            pass
        try:
            if fixed:
                rr = cg.run_fixed(fixed)
                it, hist = fixed, [rr]
            else:
                it, rr = cg.solve(on_iteration=lambda i, r: hist.append(r))
            x = cg.gather_x()
            tot, mx = cg.residual_check()
        except SystemExit as e:
            code, it, x, tot, mx = int(e.code), len(hist), None, None, None
        if rank == 0:
            q.put((code, it, hist, x, tot, mx, cg.events, (cg.use_windows, cg.interior)))
    finally:
        dist.destroy_process_group()


def run_case(world, case):
    import queue
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = None
    for _ in range(400):  # a rank that dies takes the job down at once instead of leaving the others waiting
        try:
            out = q.get(timeout=1.0)
            break
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    if out is None:
        for p in procs:
            if p.is_alive():
                p.terminate()
        pytest.fail("a rank failed or the job timed out (exit codes %s)" % [p.exitcode for p in procs])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return out


def serial(cols, rows, vals, n, mode, flip=None):
    o = OracleMatrix(CSR, mode, cols, rows, vals, n)
    if flip is not None:
        o.inject(*flip)
    it, hist, x, fatal = o.cg(rhs(n, 1))
    ev, _ = o.events()
    return it, hist, x, fatal, ev


def uneven_bounds(rows, n, world):
    """row blocks of (nearly) equal nnz"""
    counts = np.bincount(rows, minlength=n)
    cum = np.cumsum(counts)
    b = [0]
    for g in range(1, world):
        b.append(int(np.searchsorted(cum, cum[-1] * g / world)))
    return b + [n]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("matrix", ["laplace", "random"])
def test_sharded_cg_matches_single_process(world, matrix):
    cols, rows, vals, n = laplace5(24, 24) if matrix == "laplace" else random_spd(300, 8, seed=3)
    bounds = uneven_bounds(rows, n, world)
    it_s, hist_s, x_s, _, _ = serial(cols, rows, vals, n, "none")
    code, it, hist, x, tot, mx, events, windows = run_case(world, (cols, rows, vals, n, bounds, "none", None))
    assert code == 0 and it == it_s
    assert np.allclose(hist, hist_s, rtol=1e-10, atol=0)
    assert np.abs(x - x_s).max() <= 1e-10 * np.abs(x_s).max()
    assert events == []
    windows, interior = windows
    assert windows == (matrix == "laplace")  # banded: halo windows; scattered: all-gather
    # banded: all rows but the first / last grid line of the shard run beside the exchange
    if matrix == "laplace":
        r0, r1 = bounds[0], bounds[1]
        assert interior == (0, r1 - r0 - 24)  # rank 0 has no lower neighbour
    else:
        assert interior is None
    # cg.cpp:131-144 error report, against a dense recomputation
    import scipy.sparse as sp
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    err = np.abs(rhs(n, 1) - A @ x)
    assert abs(tot - np.sqrt((err * err).sum())) < 1e-12 and abs(mx - err.max()) < 1e-12


def test_sharded_secded_corrects_and_reports_global_index():
    cols, rows, vals, n = laplace5(20, 20)
    bounds = uneven_bounds(rows, n, 2)
    gi = len(vals) - 50  # lives on rank 1
    it_s, hist_s, x_s, _, ev_s = serial(cols, rows, vals, n, "secded", (gi, [13]))
    code, it, hist, x, tot, mx, events, _ = run_case(2, (cols, rows, vals, n, bounds, "secded", (gi, [13])))
    assert code == 0 and it == it_s
    assert events == ev_s == [(2, gi, 13)]
    assert np.abs(x - x_s).max() <= 1e-10 * np.abs(x_s).max()


def test_sharded_sed_fatal_stops_every_rank():
    cols, rows, vals, n = laplace5(20, 20)
    bounds = uneven_bounds(rows, n, 2)
    gi = 7  # rank 0
    code, it, hist, x, tot, mx, events, _ = run_case(2, (cols, rows, vals, n, bounds, "sed", (gi, [70])))
    assert code == 1 and it == 0
    assert events == [(1, gi, 0)]


def test_pad_columns_layout():
    from abft_sparse_cg_amd.distributed import pad_columns
    bounds = [0, 3, 10, 12]
    p, owner = pad_columns(np.array([0, 2, 3, 9, 10, 11], dtype=np.uint32), bounds, 7)
    assert list(owner) == [0, 0, 1, 1, 2, 2]
    assert list(p) == [0, 2, 7, 13, 14, 15]


@pytest.mark.parametrize("world,matrix", [(1, "random"), (2, "random"), (3, "random"), (2, "laplace"), (3, "laplace")])
def test_fixed_iteration_loop_with_device_scalars(world, matrix):
    """run_fixed (alpha, beta formed on the device, nothing read back per iteration)
    walks the same iterates as the reference loop."""
    cols, rows, vals, n = random_spd(300, 8, seed=3) if matrix == "random" else laplace5(24, 24)
    bounds = uneven_bounds(rows, n, world)
    iters = 12
    o = OracleMatrix(CSR, "none", cols, rows, vals, n)
    it_s, hist_s, x_s, _ = o.cg(rhs(n, 1), max_itrs=iters, conv=0.0)
    assert it_s == iters
    code, it, hist, x, tot, mx, events, _ = run_case(world, (cols, rows, vals, n, bounds, "none", None, iters))
    assert code == 0 and it == iters
    assert abs(hist[-1] - hist_s[-1]) <= 1e-10 * hist_s[-1]
    assert np.abs(x - x_s).max() <= 1e-10 * np.abs(x_s).max()


@pytest.mark.parametrize("fixed", [0, 9])
def test_short_exchanges_are_not_overlapped(fixed):
    """default threshold: the halo of this small banded matrix is exchanged by the blocking
    collectives and the SpMV runs as one launch -- same iterates"""
    cols, rows, vals, n = laplace5(24, 24)
    bounds = uneven_bounds(rows, n, 3)
    o = OracleMatrix(CSR, "secded", cols, rows, vals, n)
    if fixed:
        it_s, hist_s, x_s, _ = o.cg(rhs(n, 1), max_itrs=fixed, conv=0.0)
    else:
        it_s, hist_s, x_s, _ = o.cg(rhs(n, 1))
    code, it, hist, x, tot, mx, events, _ = run_case(3, (cols, rows, vals, n, bounds, "secded", None, fixed, 2 << 20))
    assert code == 0 and it == it_s and events == []
    assert abs(hist[-1] - hist_s[-1]) <= 1e-10 * hist_s[-1]
    assert np.abs(x - x_s).max() <= 1e-10 * np.abs(x_s).max()
