"""Parity at BASELINE.json's full sizes, through properties that do not need a
second implementation at that size (plus one oracle CG run, which the CPU
finishes in seconds):

  config 2/3  CSR, 5-point Laplacian 3162 x 3162 (N = 9 998 244, nnz = 49 978 572)
  config 4    CSR, random:4194304,24 (nnz ~ 104.9 M), secded
  config 5    COO, powerlaw:2097152 (nnz ~ 26 M), sec7

Properties: A.1 equals the analytic row sums bit-for-bit; scaling x by 2 scales y
by 2 bit-for-bit; CSR and COO give bit-identical y on a symmetric matrix (the
reference's two executables do); x.(A z) = z.(A x); injected
single-bit flips at full size are each reported once with the right global
index, repaired in place, and leave y bit-identical to the fault-free y."""
import ctypes

import numpy as np
import pytest

from _oracle import COO, CSR, OracleMatrix

pytestmark = pytest.mark.gpu

LAP = "laplace5:3162,3162"
NX = 3162


@pytest.fixture(scope="module")
def amd():
    import abft_sparse_cg_amd as a
    return a


@pytest.fixture(scope="module")
def gen():
    from abft_sparse_cg_amd import generators
    return generators


def bits_equal(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


class Run:
    def __init__(self, amd, fmt, mode, mat):
        cols, rows, vals, n = mat
        self.events = []
        self.ctx = amd.HIPContext(mode, fmt, on_event=lambda ev, fatal: self.events.extend(ev))
        self.A = self.ctx.create_matrix(cols, rows, vals, n, len(vals))
        self.n = n
        self.vx, self.vy = self.ctx.create_vector(n), self.ctx.create_vector(n)

    def spmv(self, x):
        self.ctx.upload(self.vx, x)
        self.ctx.spmv(self.A, self.vx, self.vy)
        return self.ctx.download(self.vy)

    def close(self):
        self.ctx.close()


def laplace_row_sums(nx, ny):
    i = np.arange(nx * ny)
    ix, iy = i % nx, i // nx
    nb = (ix > 0).astype(float) + (ix < nx - 1) + (iy > 0) + (iy < ny - 1)
    return 4.0 - nb


def test_config2_laplacian_properties_all_modes(amd, gen):
    mat = gen.generate(LAP)
    cols, rows, vals, n = mat
    assert n == 9998244 and len(vals) == 49978572
    rng = np.random.default_rng(3)
    x, z = rng.standard_normal(n), rng.standard_normal(n)
    want_ones = laplace_row_sums(NX, NX)
    y_ref = None
    for fmt, mode in (("csr", "none"), ("csr", "secded"), ("csr", "sed"), ("coo", "sec8"), ("csr", "constraints")):
        r = Run(amd, fmt, mode, mat)
        try:
            assert bits_equal(r.spmv(np.ones(n)), want_ones), (fmt, mode)
            y = r.spmv(x)
            if y_ref is None:
                y_ref = y
                # independent check of y on a slice of rows, in float64 with the same order of operations
                for row in (0, 1, NX, n // 2, n - 1):
                    lo, hi = np.searchsorted(rows, row), np.searchsorted(rows, row + 1)
                    acc = 0.0
                    for k in range(lo, hi):
                        acc += vals[k] * x[cols[k]]
                    assert y[row] == acc
            assert bits_equal(y, y_ref), (fmt, mode)  # every mode and both formats: identical bits
            assert bits_equal(r.spmv(2.0 * x), 2.0 * y)
            assert r.events == []
            if fmt == "csr" and mode == "none":
                yz = r.spmv(z)
                a, b = float(np.dot(z, y)), float(np.dot(x, yz))
                assert abs(a - b) <= 1e-11 * np.abs(z * y).sum()
        finally:
            r.close()


def test_config3_full_size_flip_round_trip(amd, gen):
    """sed detects at full size; secded repairs value, index and check-bit flips
    spread over the whole matrix, once each, and y is unchanged."""
    mat = gen.generate(LAP)
    cols, rows, vals, n = mat
    nnz = len(vals)
    x = np.random.default_rng(4).standard_normal(n)
    r = Run(amd, "csr", "secded", mat)
    try:
        y0 = r.spmv(x)
        flips = [(0, 0), (1, 63), (nnz // 3, 64), (nnz // 2, 80), (nnz - 1, 95), (12345678, 88), (nnz - 2, 37)]
        for idx, bit in flips:
            r.ctx.inject_at(r.A, idx, [bit])
        y1 = r.spmv(x)
        got = sorted(r.events)
        want = sorted((3, idx, 0) if bit == 88 else (2, idx, bit) for idx, bit in flips)
        assert got == want
        assert bits_equal(y1, y0)
        r.events.clear()
        assert bits_equal(r.spmv(x), y0) and r.events == []  # repaired in place: silent now
        r.ctx.inject_at(r.A, 7777777, [3, 70])
        r.spmv(x)
        assert r.events == [(4, 7777777, 0)]
    finally:
        r.close()
    s = Run(amd, "csr", "sed", mat)
    try:
        s.ctx.inject_at(s.A, nnz - 5, [66])
        s.spmv(x)
        assert s.events == [(1, nnz - 5, 0)]
    finally:
        s.close()


def test_config2_cg_history_matches_oracle_at_full_size(amd, gen):
    mat = gen.generate(LAP)
    cols, rows, vals, n = mat
    b = np.random.default_rng(1).random(n)
    iters = 25
    o = OracleMatrix(CSR, "none", cols, rows, vals, n)
    it_o, hist_o, x_o, _ = o.cg(b, max_itrs=iters, conv=0.0, threads=8)
    o.close()
    ctx = amd.HIPContext("none", "csr")
    try:
        A = ctx.create_matrix(cols, rows, vals, n, len(vals))
        vb, vx, vr, vp, vw = (ctx.create_vector(n) for _ in range(5))
        ctx.upload(vb, b)
        ctx.upload(vx, np.zeros(n))
        hist = []
        it, rr = amd.cg_solve(ctx, A, vb, vx, vr, vp, vw, max_itrs=iters, conv_threshold=0.0,
                              on_iteration=lambda i, r: hist.append(r))
        assert it == it_o == iters
        assert np.allclose(hist, hist_o, rtol=1e-10, atol=0)
        x = ctx.download(vx)
        assert np.abs(x - x_o).max() <= 1e-10 * np.abs(x_o).max()
    finally:
        ctx.close()


def test_config4_random_csr_secded_full_size(amd, gen):
    mat = gen.generate("random:4194304,24,1")
    cols, rows, vals, n = mat
    assert n == 1 << 22 and 100e6 < len(vals) < 110e6
    x = np.random.default_rng(6).standard_normal(n)
    r = Run(amd, "csr", "secded", mat)
    try:
        assert r.ctx.matrix_info(r.A) == ("sweep", 1)  # scattered columns over a 33 MB vector: one persistent launch
        # panels sized so that the average segment (4096 rows x 25 per row x width / n) fills two 2048-element
        # tiles less 1.5 standard deviations -- not a power of two
        from abft_sparse_cg_amd import capi
        npan, width = ctypes.c_int(), ctypes.c_int()
        capi.check(capi.load().abft_hip_matrix_panels(r.A.h, ctypes.byref(npan), ctypes.byref(width)))
        per_segment = 4096 * len(vals) / n * width.value / n
        assert 3900 < per_segment < 4096 - 64 and npan.value == -(-n // width.value) and width.value % 16 == 0
        y = r.spmv(x)
        # strictly diagonally dominant rows: A.1 = 1 + (rounding of the row sum), checked in exact order on sample rows
        for row in (0, 17, n // 2, n - 1):
            lo, hi = np.searchsorted(rows, row), np.searchsorted(rows, row + 1)
            acc = 0.0
            for k in range(lo, hi):
                acc += vals[k] * x[cols[k]]
            assert y[row] == acc
        assert bits_equal(r.spmv(0.5 * x), 0.5 * y)
        # the whole vector against the oracle (reference CSR/CPUContext.cpp:353-411, OpenMP over rows), bit for bit
        o = OracleMatrix(CSR, "secded", cols, rows, vals, n)
        assert bits_equal(y, o.spmv(x, threads=8))
        r.ctx.inject_at(r.A, len(vals) - 1, [90])
        r.ctx.inject_at(r.A, 50000000, [11])
        o.inject(len(vals) - 1, [90])
        o.inject(50000000, [11])
        y_hit = r.spmv(x)
        assert bits_equal(y_hit, y) and bits_equal(y_hit, o.spmv(x, threads=8))
        ev_o, fatal_o = o.events()
        assert sorted(r.events) == sorted(ev_o) == [(2, 50000000, 11), (2, len(vals) - 1, 90)] and not fatal_o
        # both repaired in place: a second pass reports nothing, on either side
        assert bits_equal(r.spmv(x), y) and len(r.events) == 2
        assert bits_equal(o.spmv(x, threads=8), y) and o.events()[0] == []
        o.close()
    finally:
        r.close()
    c = Run(amd, "coo", "secded", mat)
    try:
        assert bits_equal(c.spmv(x), y)  # symmetric matrix: COO sums in the same order as CSR
    finally:
        c.close()


def test_config5_powerlaw_coo_sec7_full_size(amd, gen):
    mat = gen.generate("powerlaw:2097152,2")
    cols, rows, vals, n = mat
    x = np.random.default_rng(7).standard_normal(n)
    c = Run(amd, "coo", "sec7", mat)
    k = Run(amd, "csr", "sec7", mat)
    try:
        y = c.spmv(x)
        assert bits_equal(y, k.spmv(x))
        for row in (0, 5, n - 1):
            lo, hi = np.searchsorted(rows, row), np.searchsorted(rows, row + 1)
            acc = 0.0
            for j in range(lo, hi):
                acc += vals[j] * x[cols[j]]
            assert y[row] == acc
        # the whole vector against the oracle's serial scatter (reference COO/CPUContext.cpp:250-290), bit for bit
        o = OracleMatrix(COO, "sec7", cols, rows, vals, n)
        assert bits_equal(y, o.spmv(x))
        c.ctx.inject_at(c.A, 1000003, [100])
        o.inject(1000003, [100])
        y_hit = c.spmv(x)
        assert bits_equal(y_hit, y) and bits_equal(y_hit, o.spmv(x))
        ev_o, fatal_o = o.events()
        assert c.events == ev_o == [(2, 1000003, 100)] and not fatal_o
        # a low column bit flipped where sec7 sees it: repaired before the product is placed, nothing moves
        c.ctx.inject_at(c.A, 20000001, [2])
        o.inject(20000001, [2])
        assert bits_equal(c.spmv(x), o.spmv(x)) and c.events[1:] == o.events()[0] == [(2, 20000001, 2)]
        o.close()
    finally:
        c.close()
        k.close()


def test_config5_matrix_coo_constraints_full_size(amd, gen):
    """Round 4: cg-coo -m constraints on config 5's matrix (panel layout, one paced launch).  The hot path compares every
    element with its create-time copy; a changed one runs the reference's checks (COO/CPUContext.cpp:155-188) on the
    stored words.  Whole vector against the oracle; a value-bit flip (no constraint sees it: the product changes, in both);
    index flips that keep / break the order, in neighbouring elements of the caller's order."""
    mat = gen.generate("powerlaw:2097152,2")
    cols, rows, vals, n = mat
    nnz = len(vals)
    x = np.random.default_rng(17).standard_normal(n)
    c = Run(amd, "coo", "constraints", mat)
    o = OracleMatrix(COO, "constraints", cols, rows, vals, n)
    try:
        assert c.ctx.matrix_info(c.A)[0] == "panels"
        assert bits_equal(c.spmv(x), o.spmv(x)) and c.events == [] and o.events() == ([], False)
        for index, bits in ((12345678, [77]), (nnz - 1, [64 + 52])):  # value bits
            c.ctx.inject_at(c.A, index, bits)
            o.inject(index, bits)
        assert bits_equal(c.spmv(x), o.spmv(x)) and c.events == [] and o.events() == ([], False)
        # an element whose successor in its row is far away: raising its column by one keeps the order -- the product moves
        # to the next output, as in the reference (the copy differs, the checks pass)
        k = next(i for i in range(5000000, nnz - 1) if rows[i] == rows[i + 1] and cols[i + 1] > cols[i] + 2 and not cols[i] & 1)
        c.ctx.inject_at(c.A, k, [0])
        o.inject(k, [0])
        assert bits_equal(c.spmv(x), o.spmv(x)) and c.events == [] and o.events() == ([], False)
        # ... and two neighbours changed at once, the first one out of order: one event, at the lower index, as the reference stops there
        c.ctx.inject_at(c.A, k + 1, [31])  # column beyond the matrix
        o.inject(k + 1, [31])
        c.ctx.inject_at(c.A, k, [32 + 20])  # row far up: above its successor's
        o.inject(k, [32 + 20])
        c.spmv(x)
        o.spmv(x)
        ev_o, fatal_o = o.events()
        assert fatal_o and c.events[:1] == ev_o[:1] and ev_o[0][1] in (k - 1, k)
    finally:
        c.close()
        o.close()


def test_panel_layout_on_config2_matches_streaming_layout(amd, gen, monkeypatch):
    """10 M rows = 4 883 output groups of the panel kernels (more workgroups than
    the fused-dot fold's first block): y and the fused p.w must equal the streaming
    layout's, bit for bit / to reduction tolerance."""
    mat = gen.generate(LAP)
    cols, rows, vals, n = mat
    x = np.random.default_rng(8).standard_normal(n)
    ref = Run(amd, "csr", "sec8", mat)
    try:
        y0 = ref.spmv(x)
        d0 = ref.ctx.dot(ref.vx, ref.vy)
    finally:
        ref.close()
    monkeypatch.setenv("ABFT_HIP_LAYOUT", "panels")
    pan = Run(amd, "csr", "sec8", mat)
    try:
        y1 = pan.spmv(x)
        d1 = pan.ctx.dot(pan.vx, pan.vy)  # served by the panel kernel's fused epilogue
        assert bits_equal(y1, y0)
        assert abs(d1 - d0) <= 1e-12 * float(np.abs(x * y0).sum())
        pan.ctx.inject_at(pan.A, 31415926, [9])
        assert bits_equal(pan.spmv(x), y0) and pan.events == [(2, 31415926, 9)]
    finally:
        pan.close()


def test_largest_matrix_the_ecc_modes_admit(amd, gen):
    """The ECC modes keep their check bits in the column's top byte (reference CSR/CPUContext.cpp:238, 282, 338,
    404: `& 0x00FFFFFF`), so N = 2^24 is the largest matrix they can hold: the 5-point Laplacian 4096 x 4096
    (N = 16 777 216, nnz = 83 869 696, columns up to 0xFFFFFF -- all 24 index bits set).  A.1 equals the analytic
    row sums bit for bit in secded (CSR) and sec7 (COO); a flip in the element that carries column 0xFFFFFF and
    one in its check byte are repaired and reported with the right index; one row more is refused loudly."""
    nx = 4096
    mat = gen.generate("laplace5:%d,%d" % (nx, nx))
    cols, rows, vals, n = mat
    nnz = len(vals)
    assert n == 1 << 24 and nnz == 5 * n - 4 * nx and int(cols.max()) == 0xFFFFFF
    want = laplace_row_sums(nx, nx)
    last = nnz - 1  # (row n - 1, column n - 1): the diagonal entry with every index bit set
    assert cols[last] == 0xFFFFFF and rows[last] == n - 1
    for fmt, mode in (("csr", "secded"), ("coo", "sec7")):
        r = Run(amd, fmt, mode, mat)
        try:
            assert bits_equal(r.spmv(np.ones(n)), want), (fmt, mode)
            assert r.events == []
            bits = [64 + 23, 64 + 30] if fmt == "csr" else [23, 30]  # the top index bit, a check bit
            for b in bits:
                r.ctx.inject_at(r.A, last, [b])
                assert bits_equal(r.spmv(np.ones(n)), want)
                assert r.events == [(2, last, b)], (fmt, b, r.events)
                r.events.clear()
            assert bits_equal(r.spmv(np.ones(n)), want) and r.events == []
        finally:
            r.close()
    big = gen.generate("laplace5:%d,%d" % (nx, nx + 1))  # N = 2^24 + 4096: columns need a 25th bit
    ctx = amd.HIPContext("secded", "csr", on_event=lambda e, f: None)
    try:
        with pytest.raises(amd.AbftError) as e:
            ctx.create_matrix(big[0], big[1], big[2], big[3], len(big[2]))
        assert e.value.code == -4 and "24 bits" in str(e.value)
    finally:
        ctx.close()
    ctx = amd.HIPContext("none", "csr", on_event=lambda e, f: None)  # (no such limit without check bits)
    try:
        A = ctx.create_matrix(big[0], big[1], big[2], big[3], len(big[2]))
        vx, vy = ctx.create_vector(big[3]), ctx.create_vector(big[3])
        ctx.upload(vx, np.ones(big[3]))
        ctx.spmv(A, vx, vy)
        assert bits_equal(ctx.download(vy), laplace_row_sums(nx, nx + 1))
    finally:
        ctx.close()


def test_constraints_mode_full_size_on_the_sweep_layout(amd, gen, monkeypatch):
    """Round 3: config 4's matrix in constraints mode now takes the sweep layout (round 2: streaming, 2.5 x slower).
    Against the streaming layout of the same matrix: y bit for bit; an index flip that breaks the column order
    inside a panel, one that breaks it across a panel boundary (the row's previous element sits in another
    panel: the register-carried check) and one that leaves the vector are each reported as the streaming
    layout -- the reference's row-by-row loop -- reports them (CSR/CPUContext.cpp:186-200)."""
    mat = gen.generate("random:4194304,24,1")
    cols, rows, vals, n = mat
    x = np.random.default_rng(21).standard_normal(n)
    # elements to corrupt: the first element of a row's second half (its predecessor is likely in another panel)
    # and a mid-row element; bit 64 + k = bit k of the column word
    r0 = 1234567
    lo, hi = int(np.searchsorted(rows, r0)), int(np.searchsorted(rows, r0 + 1))
    assert hi - lo >= 12
    cases = [(lo + (hi - lo) // 2, 64 + 21), (lo + 3, 64 + 0), (lo + 5, 64 + 30), (hi - 1, 64 + 22)]
    results = {}
    for layout in ("stream", "sweep"):
        monkeypatch.setenv("ABFT_HIP_LAYOUT", layout)
        r = Run(amd, "csr", "constraints", mat)
        try:
            assert r.ctx.matrix_info(r.A)[0] == layout
            y = r.spmv(x)
            assert r.events == []
            out = [y]
            for idx, bit in cases:
                r.ctx.inject_at(r.A, idx, [bit])
                r.spmv(x)
                out.append(list(r.events))
                r.events.clear()
                r.ctx.inject_at(r.A, idx, [bit])  # flip it back
                assert bits_equal(r.spmv(x), y) and r.events == []
            results[layout] = out
        finally:
            r.close()
    assert bits_equal(results["stream"][0], results["sweep"][0])
    assert results["stream"][1:] == results["sweep"][1:], (results["stream"][1:], results["sweep"][1:])
    kinds = {ev[0][0] for ev in results["sweep"][1:] if ev}
    assert 8 in kinds and len([ev for ev in results["sweep"][1:] if ev]) >= 3  # column-order violations among them


def test_config4_row_partitioned_shards_equal_the_one_gpu_product(amd, gen):
    """BASELINE.json configs[3] in its stated form -- cg-csr -m secded row-partitioned over 8 ranks -- shard by
    shard on the one GPU: for ranks 0, 3 and 7 the shard is planned by the product's planner
    (host/partition.cpp: abft_plan_shard), its row block generated alone (abft_gen_fill: what a rank's host does,
    CGContextExt::create_matrix_rows), created with abft_hip_matrix_create_shard (gather indices re-based to the
    slot-padded gathered vector, events carrying global element indices) and multiplied against the gathered
    vector: the slice of the one-GPU y bit for bit, before and after two injected flips, which are reported
    with their GLOBAL index and repaired (SURVEY 8e 'ECC/event semantics'; reference
    CSR/CPUContext.cpp:353-411 per shard)."""
    import ctypes as C
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _partition_worker import plan
    spec, G = "random:4194304,24,1", 8
    mat = gen.generate(spec)
    cols, rows, vals, n = mat
    x = np.random.default_rng(12).standard_normal(n)
    one = Run(amd, "csr", "secded", mat)
    try:
        y_full = one.spmv(x)
        assert one.events == []
    finally:
        one.close()
    L = C.CDLL(gen.LIB_PATH)
    block_bounds = gen.partition(spec, G)  # the cut every rank's host makes without holding the matrix
    for k in (0, 3, 7):
        p = plan(L, 0, cols, rows, vals, n, G, k)
        assert [int(b) for b in p["bounds"]] == block_bounds
        b0, b1 = block_bounds[k], block_bounds[k + 1]
        first, cnt = p["first"], p["nnz"]
        assert (p["out0"], p["n_loc"]) == (b0, b1 - b0) and p["n_pad"] == G * p["slot"]
        bc, br, bv, _ = gen.generate(spec, b0, b1)  # this rank's rows only
        assert len(bv) == cnt and np.array_equal(bc, cols[first:first + cnt]) and np.array_equal(br, rows[first:first + cnt])
        assert np.array_equal(bv.view(np.uint64), vals[first:first + cnt].view(np.uint64))
        assert np.array_equal(p["lout"], br - b0)
        # an eighth of the elements each (the cut is by non-zeros), every other rank's slot read (scattered columns)
        assert abs(cnt - len(vals) / G) < 1e-3 * len(vals) and p["interior"] == (0, 0)
        assert all(hi - lo > 0.99 * (block_bounds[g + 1] - block_bounds[g]) for g, (lo, hi) in enumerate(p["need"]) if g != k)
        events = []
        ctx = amd.HIPContext("secded", "csr", on_event=lambda ev, fatal: events.extend(ev))
        try:
            A = ctx.create_matrix(p["pin"], p["lout"], bv, p["n_loc"], cnt, n_in=p["n_pad"], index_base=first)
            # 524 288 rows against the 33.5 MB gathered vector: the sweep layout, the smallest row groups
            # (2 rows per thread: 1 024 workgroups), 2 MB panels (the segments would stay under a tile at 1 MB)
            assert ctx.matrix_info(A) == ("sweep", 1)
            npan, width = ctypes.c_int(), ctypes.c_int()
            from abft_sparse_cg_amd import capi
            capi.check(capi.load().abft_hip_matrix_panels(A.h, ctypes.byref(npan), ctypes.byref(width)))
            assert 1 << 18 <= width.value <= 5 << 16 and npan.value == -(-p["n_pad"] // width.value)
            xpad = np.zeros(p["n_pad"])
            for g in range(G):
                xpad[g * p["slot"]:g * p["slot"] + block_bounds[g + 1] - block_bounds[g]] = x[block_bounds[g]:block_bounds[g + 1]]
            vx, vy = ctx.create_vector(p["n_pad"]), ctx.create_vector(p["n_loc"])
            ctx.upload(vx, xpad)
            ctx.spmv(A, vx, vy)
            want = y_full[b0:b1]
            assert bits_equal(ctx.download(vy), want), k
            ctx._drain()
            assert events == []
            flips = [(cnt // 5, 3), (cnt - 7, 70 + k)]  # a value bit and a column / check bit, local element indices
            for i, bit in flips:
                ctx.inject_at(A, i, [bit])
            ctx.spmv(A, vx, vy)
            assert bits_equal(ctx.download(vy), want), k
            ctx._drain()
            assert sorted(events) == sorted((2, first + i, bit) for i, bit in flips), (k, events)
            events.clear()
            ctx.spmv(A, vx, vy)  # repaired in place: silent now
            assert bits_equal(ctx.download(vy), want)
            ctx._drain()
            assert events == []
        finally:
            ctx.close()


@pytest.mark.parametrize("spec,rpt", [("random:4194304,24,1", "2"), ("random:4194304,24,1", "16"), (LAP, "16"), (LAP, "8")])
def test_sweep_layout_full_size_matches_streaming_layout(amd, gen, monkeypatch, spec, rpt):
    """The sweep layout at full size against the streaming layout of the same matrix, every row bit
    for bit, the fused p.w to reduction tolerance, a flip reported with the caller's index: config 4
    with the smallest and the largest group size (8 192 groups in several rounds / 1 024 in one),
    and forced onto config 2's 10 M rows (2 442 or 4 883 groups: more than are resident at once)."""
    mat = gen.generate(spec)
    cols, rows, vals, n = mat
    x = np.random.default_rng(9).standard_normal(n)
    monkeypatch.setenv("ABFT_HIP_LAYOUT", "stream")
    ref = Run(amd, "csr", "sec7", mat)
    try:
        y0 = ref.spmv(x)
        d0 = ref.ctx.dot(ref.vx, ref.vy)
    finally:
        ref.close()
    monkeypatch.setenv("ABFT_HIP_LAYOUT", "sweep")
    monkeypatch.setenv("ABFT_HIP_SWEEP_RPT", rpt)
    sw = Run(amd, "csr", "sec7", mat)
    try:
        assert sw.ctx.matrix_info(sw.A)[0] == "sweep"
        y1 = sw.spmv(x)
        d1 = sw.ctx.dot(sw.vx, sw.vy)
        assert bits_equal(y1, y0)
        assert abs(d1 - d0) <= 1e-12 * float(np.abs(x * y0).sum())
        sw.ctx.inject_at(sw.A, len(vals) // 3, [70])
        assert bits_equal(sw.spmv(x), y0) and sw.events == [(2, len(vals) // 3, 70)]
        assert bits_equal(sw.spmv(x), y0)  # again: the pacing board was reset by the last workgroup
    finally:
        sw.close()
