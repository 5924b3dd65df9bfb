"""Parity of the HIP path (through the C ABI, libabft_hip.so) against the CPU
oracle on the same seeded inputs and against the committed golden vectors.

Bars (BASELINE.json north_star):
  * stored (ECC-encoded) elements, SpMV results, element-wise results of
    calc_xr / calc_p, ECC / constraint event streams: BIT-EXACT;
  * the two reductions (dot, calc_xr's r.r) are tree sums on the GPU and serial
    sums in the reference: relative 1e-13 here (target: residual within 1e-10);
  * CG: identical iteration count, rr history within 1e-10 relative.
"""
import ctypes
import json
import os

import numpy as np
import pytest

from _oracle import (COO, CSR, MODES, OracleMatrix, event_lines, laplace5, ora_calc_p, ora_calc_xr, ora_dot,
                     random_spd, rhs)

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FMTS = [CSR, COO]
FNAME = {CSR: "csr", COO: "coo"}
NBITS = {CSR: 96, COO: 128}


@pytest.fixture(scope="module")
def amd():
    import abft_sparse_cg_amd as a
    return a


class Hip:
    """One HIPContext + matrix + work vectors, events collected not printed."""

    def __init__(self, amd, fmt, mode, cols, rows, vals, n, n_in=None, index_base=0):
        self.events, self.fatal = [], False
        self.ctx = amd.HIPContext(mode, FNAME[fmt], on_event=self._on)
        self.fmt, self.n, self.n_in = fmt, n, n if n_in is None else n_in
        self.A = self.ctx.create_matrix(cols, rows, vals, n, len(vals), n_in=n_in, index_base=index_base)
        self.vx = self.ctx.create_vector(self.n_in)
        self.vy = self.ctx.create_vector(n)

    def _on(self, ev, fatal):
        self.events += ev
        self.fatal |= fatal

    def spmv(self, x):
        self.ctx.upload(self.vx, x)
        self.ctx.upload(self.vy, np.full(self.n, np.nan))
        self.ctx.spmv(self.A, self.vx, self.vy)
        return self.ctx.download(self.vy)

    def take_events(self):
        ev, f = self.events, self.fatal
        self.events, self.fatal = [], False
        return ev, f

    def close(self):
        self.ctx.destroy_vector(self.vx)
        self.ctx.destroy_vector(self.vy)
        self.ctx.destroy_matrix(self.A)
        self.ctx.close()


def bits_equal(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


def ragged(n, seed, long_rows=()):
    """Irregular square matrix: empty rows, 1-entry rows, a few rows longer than
    one LDS tile (1024 nnz) and than several tiles; sorted by (row, col)."""
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for r in range(n):
        k = int(rng.choice([0, 0, 1, 2, 3, 7, 20, 60]))
        if r in long_rows:
            k = long_rows[r]
        k = min(k, n)
        c = np.sort(rng.choice(n, size=k, replace=False))
        rows.append(np.full(k, r))
        cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = rng.standard_normal(len(rows)) * 10.0 ** rng.integers(-3, 4, size=len(rows))
    return cols.astype(np.uint32), rows.astype(np.uint32), vals, n


MATS = {
    "lap9x7": lambda: laplace5(9, 7),
    "lap40": lambda: laplace5(40, 33),
    "rnd300": lambda: random_spd(300, 10, seed=5),
    "ragged": lambda: ragged(700, 1, {5: 1500, 6: 3, 300: 5000, 699: 1025}),
    "one": lambda: (np.array([0], np.uint32), np.array([0], np.uint32), np.array([2.5]), 1),
    "empty": lambda: (np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0), 5),
    "tail_empty": lambda: (np.array([0, 1], np.uint32), np.array([0, 0], np.uint32), np.array([1.0, 2.0]), 2000),
}


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("mat", sorted(MATS))
def test_spmv_and_encoding_bit_exact(amd, fmt, mode, mat):
    cols, rows, vals, n = MATS[mat]()
    x = rhs(n, 11) - 0.5
    o = OracleMatrix(fmt, mode, cols, rows, vals, n)
    h = Hip(amd, fmt, mode, cols, rows, vals, n)
    try:
        assert np.array_equal(h.ctx.stored_words(h.A), o.stored_words())
        if fmt == CSR:
            assert np.array_equal(h.ctx.rowptr(h.A), o.csr_arrays()[1])
        y = h.spmv(x)
        want = o.spmv(x)
        ev, fatal = h.take_events()
        if mat == "ragged" and mode == "constraints":
            # unordered duplicates never occur, but equal neighbours are legal input here: same verdict
            oev, ofatal = o.events()
            assert (ev, fatal) == (oev, ofatal)
            if fatal:
                return
        else:
            assert ev == [] and not fatal
        assert bits_equal(y, want)
    finally:
        h.close()


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("mode", ["sed", "sec7", "sec8", "secded"])
def test_every_single_bit_flip(amd, fmt, mode):
    """All 96 / 128 positions of one element: same event line, same repaired
    element, same y on this pass and the next (write-back persists)."""
    cols, rows, vals, n = random_spd(64, 6, seed=9)
    x = rhs(n, 2) + 0.25
    index = len(vals) // 2
    h = Hip(amd, fmt, mode, cols, rows, vals, n)
    clean = h.ctx.stored_words(h.A)[index].copy()
    try:
        for bit in range(NBITS[fmt]):
            o = OracleMatrix(fmt, mode, cols, rows, vals, n)
            o.inject(index, [bit])
            h.ctx.inject_at(h.A, index, [bit])
            assert np.array_equal(h.ctx.stored_words(h.A)[index], o.stored_words()[index])
            y1 = h.spmv(x)
            ev, fatal = h.take_events()
            w1 = o.spmv(x)
            oev, ofatal = o.events()
            assert (ev, fatal) == (oev, ofatal), bit
            assert event_lines(ev, fmt) == event_lines(oev, fmt)
            if fatal:
                h.ctx.inject_at(h.A, index, [bit])  # undo: sed cannot repair
            else:
                assert bits_equal(y1, w1), bit
                y2, w2 = h.spmv(x), o.spmv(x)
                assert h.take_events() == o.events()
                assert bits_equal(y2, w2), bit
                assert np.array_equal(h.ctx.stored_words(h.A)[index], o.stored_words()[index]), bit
                if mode == "sec7" and bit == (88 if fmt == CSR else 24):
                    h.ctx.inject_at(h.A, index, [bit])  # sec7's blind spot: undo by hand
            assert np.array_equal(h.ctx.stored_words(h.A)[index], clean), bit
    finally:
        h.close()


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("mode", ["sec7", "sec8", "secded"])
def test_double_bit_flips(amd, fmt, mode):
    cols, rows, vals, n = random_spd(64, 6, seed=10)
    x = rhs(n, 3) + 0.25
    rng = np.random.default_rng(1)
    for _ in range(40):
        index = int(rng.integers(0, len(vals)))
        b1, b2 = (int(b) for b in rng.choice(NBITS[fmt], size=2, replace=False))
        o = OracleMatrix(fmt, mode, cols, rows, vals, n)
        o.inject(index, [b1, b2])
        h = Hip(amd, fmt, mode, cols, rows, vals, n)
        try:
            h.ctx.inject_at(h.A, index, [b1, b2])
            y, want = h.spmv(x), o.spmv(x)
            ev, fatal = h.take_events()
            assert (ev, fatal) == o.events(), (index, b1, b2)
            if mode == "secded":
                assert fatal and event_lines(ev, fmt) == ["[ECC] double-bit error detected\n"]
            elif not fatal:
                assert bits_equal(y, want), (index, b1, b2)
                assert np.array_equal(h.ctx.stored_words(h.A), o.stored_words())
        finally:
            h.close()


@pytest.mark.parametrize("layout", [None, "panels"])
@pytest.mark.parametrize("fmt", FMTS)
def test_constraints_mode_detects_like_reference(amd, fmt, layout, monkeypatch):
    """every index-bit flip of four elements: the oracle's events and, where the flip passes the checks, its y.
    `panels`: the COO panel layout forced onto the small matrix (round 3: constraints mode runs there too -- the
    check reaches an element's caller-order successor through the table of stored positions, whatever the
    layout; CSR keeps the streaming layout under this setting, its scattered-matrix form is the sweep layout:
    test_constraints_mode_in_the_sweep_layout)"""
    if layout:
        monkeypatch.setenv("ABFT_HIP_LAYOUT", layout)
        monkeypatch.setenv("ABFT_HIP_PANEL_WIDTH", "16")
        monkeypatch.setenv("ABFT_HIP_PANEL_CHUNK", "2")
    cols, rows, vals, n = random_spd(50, 6, seed=4)
    x = rhs(n, 5)
    idx_bits = range(64, 96) if fmt == CSR else range(0, 64)
    fatal_seen = 0
    for index in (0, 5, len(vals) - 1, len(vals) // 3):
        for bit in idx_bits:
            o = OracleMatrix(fmt, "constraints", cols, rows, vals, n)
            o.inject(index, [bit])
            h = Hip(amd, fmt, "constraints", cols, rows, vals, n)
            try:
                assert h.ctx.matrix_info(h.A)[0] == ("panels" if layout and fmt == COO else "stream")
                h.ctx.inject_at(h.A, index, [bit])
                y, want = h.spmv(x), o.spmv(x)
                ev, fatal = h.take_events()
                assert (ev, fatal) == o.events(), (index, bit)
                fatal_seen += fatal
                if not fatal:
                    assert bits_equal(y, want), (index, bit)
            finally:
                h.close()
    assert fatal_seen > 20


@pytest.mark.parametrize("layout", ["stream", "panels"])
def test_coo_constraints_pairs_with_changed_members(amd, layout, monkeypatch):
    """Round 4: the COO constraints check no longer reads every element's caller-order successor (one HBM gather
    per element) -- an element is compared with the copy of its {col,row} taken at create time, and only one that
    differs runs the reference's checks (COO/CPUContext.cpp:155-188), for both pairs it is a member of.  What
    that has to get right: flips in NEIGHBOURING elements of the caller's order (a pair both of whose members
    changed is checked once, by the first), a changed last / first element, a flip undone again, and a matrix
    that violates the order as created (no flips at all).  Events as the oracle's; y where nothing is fatal."""
    monkeypatch.setenv("ABFT_HIP_LAYOUT", layout)
    monkeypatch.setenv("ABFT_HIP_PANEL_WIDTH", "16")
    monkeypatch.setenv("ABFT_HIP_PANEL_CHUNK", "2")
    cols, rows, vals, n = random_spd(60, 7, seed=9)
    nnz = len(vals)
    x = rhs(n, 3)
    rng = np.random.default_rng(5)

    def compare(o, h, tag):
        fatal = False
        for _ in range(2):
            y, want = h.spmv(x), o.spmv(x)
            ev, fatal = h.take_events()
            assert (ev, fatal) == o.events(), tag
            if fatal:
                break
            assert bits_equal(y, want), tag
        return fatal

    fatal_seen = passed = 0
    for trial in range(120):
        o = OracleMatrix(COO, "constraints", cols, rows, vals, n)
        h = Hip(amd, COO, "constraints", cols, rows, vals, n)
        try:
            k = int(rng.choice([0, nnz - 3, int(rng.integers(0, nnz - 3))]))
            members = sorted(set(int(k + d) for d in rng.choice(3, size=int(rng.integers(1, 4)))))
            flips = []
            for i in members:
                word = int(rng.integers(0, 2))  # the column word or the row word
                flips.append((i, [32 * word + int(rng.integers(0, 5))]))
            for i, bits in flips:
                o.inject(i, bits)
                h.ctx.inject_at(h.A, i, bits)
            f = compare(o, h, (trial, flips))
            fatal_seen += f
            passed += not f
            if not f and trial % 3 == 0:  # ... and undone: nothing differs from its copy any more
                for i, bits in flips:
                    o.inject(i, bits)
                    h.ctx.inject_at(h.A, i, bits)
                assert not compare(o, h, (trial, "undone"))
        finally:
            h.close()
    assert fatal_seen >= 30 and passed >= 10, (fatal_seen, passed)
    # as created: two neighbours of the caller's order exchanged (the first pair check fails on every pass, no flip)
    for k in (0, nnz // 2, nnz - 2):
        c2, r2, v2 = cols.copy(), rows.copy(), vals.copy()
        for a in (c2, r2, v2):
            a[[k, k + 1]] = a[[k + 1, k]]
        o = OracleMatrix(COO, "constraints", c2, r2, v2, n)
        h = Hip(amd, COO, "constraints", c2, r2, v2, n)
        try:
            assert compare(o, h, ("created", k))
            assert compare(o, h, ("created again", k))
        finally:
            h.close()


@pytest.mark.parametrize("width,rpt,lag", [(16, 8, 2), (7, 2, 0), (64, 4, 1), (100000, 8, 2)])
def test_constraints_mode_in_the_sweep_layout(amd, width, rpt, lag, monkeypatch):
    """Round 3: constraints mode runs on the sweep layout too (scattered matrices; round 2 forced the
    streaming layout there).  The reference's checks (CSR/CPUContext.cpp:173-200) are made in the summing
    phase in the row's order -- a row's elements sit in several panels, so the order check between the last
    element of one panel and the first of the next goes through a register per row.  Forced here with tiny
    panels: every index-bit flip of several elements -> the oracle's events (kind, caller's index) and, when
    the flip passes the checks, the oracle's y bit for bit; then the two row-pointer checks."""
    from abft_sparse_cg_amd import capi
    monkeypatch.setenv("ABFT_HIP_LAYOUT", "sweep")
    monkeypatch.setenv("ABFT_HIP_PANEL_WIDTH", str(width))
    monkeypatch.setenv("ABFT_HIP_SWEEP_RPT", str(rpt))
    monkeypatch.setenv("ABFT_HIP_SWEEP_LAG", str(lag))
    cols, rows, vals, n = random_spd(50, 6, seed=4)
    x = rhs(n, 5)
    fatal_seen = passed = 0
    kinds = set()
    for index in (0, 5, len(vals) - 1, len(vals) // 3, len(vals) // 2 + 1):
        for bit in range(64, 96):
            o = OracleMatrix(CSR, "constraints", cols, rows, vals, n)
            o.inject(index, [bit])
            h = Hip(amd, CSR, "constraints", cols, rows, vals, n)
            try:
                assert h.ctx.matrix_info(h.A)[0] == "sweep"
                h.ctx.inject_at(h.A, index, [bit])
                y, want = h.spmv(x), o.spmv(x)
                ev, fatal = h.take_events()
                assert (ev, fatal) == o.events(), (index, bit, ev, o.events())
                fatal_seen += fatal
                if fatal:
                    kinds.add(ev[0][0])
                else:
                    passed += 1
                    assert bits_equal(y, want), (index, bit)
            finally:
                h.close()
    assert fatal_seen > 20 and passed > 5 and {7, 8} <= kinds  # column size and column order both seen
    # the clean matrix, twice (fused dot on the second pass), on the ragged one too
    for mat in ("ragged", "rnd300", "tail_empty"):
        c2, r2, v2, n2 = MATS[mat]()
        x2 = rhs(n2, 11) - 0.5
        o = OracleMatrix(CSR, "constraints", c2, r2, v2, n2)
        h = Hip(amd, CSR, "constraints", c2, r2, v2, n2)
        try:
            if mat != "ragged" or width <= 64:
                assert h.ctx.matrix_info(h.A)[0] == "sweep", mat
            assert bits_equal(h.spmv(x2), o.spmv(x2)) and h.take_events() == ([], False)
            h.ctx.spmv(h.A, h.vx, h.vy)
            d = h.ctx.dot(h.vx, h.vy)
            yy = h.ctx.download(h.vy)
            assert bits_equal(yy, o.spmv(x2)) and abs(d - ora_dot(x2, yy)) <= 1e-13 * float(np.abs(x2 * yy).sum()) + 1e-300
        finally:
            h.close()
    # row pointers: the reference's two checks, same first fatal event
    cols, rows, vals, n = random_spd(120, 6, seed=8)
    x = rhs(n, 5)
    rk = set()
    for row, mask in ((7, 1 << 30), (7, 1 << 3), (1, 1 << 5), (n, 1 << 29), (60, 1 << 9), (119, 1 << 1), (33, 1 << 4)):
        o = OracleMatrix(CSR, "constraints", cols, rows, vals, n)
        o._view("ora_matrix_csr_rowptr", np.uint32, n + 1)[row] ^= np.uint32(mask)
        h = Hip(amd, CSR, "constraints", cols, rows, vals, n)
        try:
            assert h.ctx.matrix_info(h.A)[0] == "sweep"
            capi.check(h.ctx.L.abft_hip_inject_rowptr(h.A.h, row, mask))
            h.spmv(x)
            o.spmv(x)
            ev, fatal = h.take_events()
            oev, ofatal = o.events()
            if ofatal and oev[0][0] in (5, 6):  # (a pointer moved without breaking either check: see DESIGN.md)
                assert fatal and ev[:1] == oev[:1], (row, mask, ev, oev)
                rk.add(ev[0][0])
        finally:
            h.close()
    assert {5, 6} <= rk


def test_event_queue_overflow_is_reported_not_hidden(amd):
    """More events than the device queue holds (65 536): the drain hands over what was kept and
    fails loudly (ABFT_ERR_RANGE) -- a run past this point would no longer print the reference's
    lines, and dropping the surplus silently could hide a fatal one."""
    from abft_sparse_cg_amd import capi
    cols, rows, vals, n = laplace5(130, 130)
    assert len(vals) > 70000
    ctx = amd.HIPContext("sec7", "csr", on_event=lambda ev, fatal: None)
    try:
        A = ctx.create_matrix(cols, rows, vals, n, len(vals))
        for i in range(66000):
            ctx.inject_at(A, i, [3])
        vx, vy = ctx.create_vector(n), ctx.create_vector(n)
        ctx.upload(vx, rhs(n, 1))
        capi.check(ctx.L.abft_hip_spmv(ctx.h, A.h, vx.h, vy.h))
        with pytest.raises(amd.AbftError) as e:
            ctx.drain_events()
        assert e.value.code == -4 and "overflow" in str(e.value) and "66000" in str(e.value)
        # the queue is empty again and the repaired matrix multiplies cleanly
        capi.check(ctx.L.abft_hip_spmv(ctx.h, A.h, vx.h, vy.h))
        assert ctx.drain_events() == ([], False)
        assert bits_equal(ctx.download(vy), OracleMatrix(CSR, "sec7", cols, rows, vals, n).spmv(rhs(n, 1)))
    finally:
        ctx.close()


def test_constraints_mode_row_pointer_checks(amd):
    """The two checks no element flip reaches (reference CSR/CPUContext.cpp:173-182: "row size" when
    a row's end lies past nnz, "row order" when it lies before its start): a bit of a row pointer
    flipped on both sides -- same first fatal event as the oracle's run over the same arrays."""
    from abft_sparse_cg_amd import capi
    cols, rows, vals, n = random_spd(120, 6, seed=8)
    x = rhs(n, 5)
    kinds = set()
    for row, mask in ((7, 1 << 30), (7, 1 << 3), (1, 1 << 5), (n, 1 << 29), (60, 1 << 9), (119, 1 << 1), (33, 1 << 4)):
        o = OracleMatrix(CSR, "constraints", cols, rows, vals, n)
        o._view("ora_matrix_csr_rowptr", np.uint32, n + 1)[row] ^= np.uint32(mask)
        h = Hip(amd, CSR, "constraints", cols, rows, vals, n)
        try:
            capi.check(h.ctx.L.abft_hip_inject_rowptr(h.A.h, row, mask))
            assert np.array_equal(h.ctx.rowptr(h.A), o.csr_arrays()[1])
            h.spmv(x)
            o.spmv(x)
            ev, fatal = h.take_events()
            oev, ofatal = o.events()
            assert fatal == ofatal, (row, mask)
            if fatal:
                assert ev[:1] == oev[:1], (row, mask, ev, oev)
                assert event_lines(ev[:1], CSR) == event_lines(oev[:1], CSR)
                kinds.add(ev[0][0])
        finally:
            h.close()
    assert {5, 6} <= kinds  # ABFT_EV_ROW_SIZE and ABFT_EV_ROW_ORDER both seen


def test_shard_geometry_and_global_event_index(amd):
    """Row-block shard: n_out local rows, columns index a longer vector, events
    carry index_base + local element index."""
    cols, rows, vals, n = random_spd(200, 8, seed=6)
    r0, r1 = 60, 130
    m = (rows >= r0) & (rows < r1)
    base = int(np.argmax(m))
    x = rhs(n, 8)
    for fmt in (CSR,):
        o = OracleMatrix(fmt, "secded", cols[m], rows[m] - r0, vals[m], r1 - r0, n_in=n, index_base=base)
        h = Hip(amd, fmt, "secded", cols[m], rows[m] - r0, vals[m], r1 - r0, n_in=n, index_base=base)
        try:
            o.inject(17, [3])
            h.ctx.inject_at(h.A, 17, [3])
            y, want = h.spmv(x), o.spmv(x)
            ev, fatal = h.take_events()
            assert (ev, fatal) == o.events() and ev[0][1] == base + 17
            assert bits_equal(y, want)
        finally:
            h.close()


def test_create_rejects_bad_input(amd):
    ctx = amd.HIPContext("sed", "csr", on_event=lambda e, f: None)
    try:
        one = np.array([1.0])
        with pytest.raises(amd.AbftError) as e:  # column needs more than 24 bits under ECC
            ctx.create_matrix(np.array([1 << 24], np.uint32), np.array([0], np.uint32), one, 1 << 25, 1)
        assert e.value.code == -4
        with pytest.raises(amd.AbftError):  # row outside the matrix
            ctx.create_matrix(np.array([0], np.uint32), np.array([7], np.uint32), one, 4, 1)
        with pytest.raises(amd.AbftError):  # rows not sorted
            ctx.create_matrix(np.array([0, 0], np.uint32), np.array([1, 0], np.uint32), np.array([1.0, 1.0]), 4, 2)
        A = ctx.create_matrix(np.array([0], np.uint32), np.array([0], np.uint32), one, 4, 1)
        v3, v4 = ctx.create_vector(3), ctx.create_vector(4)
        with pytest.raises(amd.AbftError):  # vector shorter than the matrix: refused on the host
            ctx.spmv(A, v3, v4)
        with pytest.raises(amd.AbftError):
            ctx.inject_at(A, 5, [1])
    finally:
        ctx.close()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 255, 1000, 4097, 1 << 20, (1 << 20) + 3])
def test_vector_kernels(amd, n):
    ctx = amd.HIPContext("none", "csr")
    rng = np.random.default_rng(n)
    try:
        a, b, p, w = (rng.standard_normal(n) for _ in range(4))
        va, vb, vp, vw = (ctx.create_vector(n) for _ in range(4))
        for v, arr in ((va, a), (vb, b), (vp, p), (vw, w)):
            ctx.upload(v, arr)
        d = ctx.dot(va, vb)
        ref = ora_dot(a, b)
        scale = float(np.abs(a * b).sum())
        assert abs(d - ref) <= 1e-13 * scale
        x_ref, r_ref = a.copy(), b.copy()
        rr_ref = ora_calc_xr(x_ref, r_ref, p, w, 0.37)
        rr = ctx.calc_xr(va, vb, vp, vw, 0.37)
        assert abs(rr - rr_ref) <= 1e-13 * float((r_ref * r_ref).sum())
        assert bits_equal(ctx.download(va), x_ref) and bits_equal(ctx.download(vb), r_ref)
        p_ref = p.copy()
        ora_calc_p(p_ref, r_ref, 1.7)
        ctx.calc_p(vp, vb, 1.7)
        assert bits_equal(ctx.download(vp), p_ref)
        ctx.copy_vector(vw, vp)
        assert bits_equal(ctx.download(vw), p_ref)
        # same call twice: reductions are deterministic
        assert ctx.dot(va, vb) == ctx.dot(va, vb)
        if n > 8:  # an odd-offset window: the unaligned (scalar) kernels
            wa, wb = ctx.view_vector(va, 1, n - 3), ctx.view_vector(vb, 1, n - 3)
            d2 = ctx.dot(wa, wb)
            assert abs(d2 - ora_dot(x_ref[1:n - 2], r_ref[1:n - 2])) <= 1e-13 * float(np.abs(x_ref * r_ref).sum())
            ctx.calc_p(wa, wb, -0.5)
            e = x_ref.copy()
            ora_calc_p(e[1:n - 2], r_ref[1:n - 2], -0.5)
            assert bits_equal(ctx.download(va), e)
    finally:
        ctx.close()


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("mode", MODES)
def test_cg_iteration_count_and_residual_history(amd, fmt, mode):
    cols, rows, vals, n = laplace5(64, 64)
    b = rhs(n, 1)
    o = OracleMatrix(fmt, mode, cols, rows, vals, n)
    it_o, hist_o, x_o, _ = o.cg(b)
    ctx = amd.HIPContext(mode, FNAME[fmt])
    try:
        A = ctx.create_matrix(cols, rows, vals, n, len(vals))
        vb, vx, vr, vp, vw = (ctx.create_vector(n) for _ in range(5))
        ctx.upload(vb, b)
        ctx.upload(vx, np.zeros(n))
        hist = []
        it, rr = amd.cg_solve(ctx, A, vb, vx, vr, vp, vw, on_iteration=lambda i, r: hist.append(r))
        assert it == it_o and it > 50
        assert np.allclose(hist, hist_o, rtol=1e-10, atol=0)
        x = ctx.download(vx)
        assert np.abs(x - x_o).max() <= 1e-10 * np.abs(x_o).max()
    finally:
        ctx.close()


def test_golden_vectors(amd):
    """The reference's own outputs (tests/golden/kernels.npz, flips.json)."""
    kern = np.load(os.path.join(G, "kernels.npz"))
    for mat in ("lap9x7", "rnd80", "lap16"):
        cols, rows, vals = kern[mat + "_cols"], kern[mat + "_rows"], kern[mat + "_vals"]
        n = int(kern[mat + "_n"][0])
        for fmt in FMTS:
            for mode in MODES:
                key = "%s_%s_%s" % (mat, FNAME[fmt], mode)
                h = Hip(amd, fmt, mode, cols, rows, vals, n)
                try:
                    assert np.array_equal(h.ctx.stored_words(h.A), kern[key + "_words"]), key
                    assert bits_equal(h.spmv(kern[mat + "_x"]), kern[key + "_y"]), key
                finally:
                    h.close()
    cases = json.load(open(os.path.join(G, "flips.json")))
    for c in cases:
        mat, fmt = c["matrix"], {"csr": CSR, "coo": COO}[c["fmt"]]
        n = int(kern[mat + "_n"][0])
        h = Hip(amd, fmt, c["mode"], kern[mat + "_cols"], kern[mat + "_rows"], kern[mat + "_vals"], n)
        try:
            h.ctx.inject_at(h.A, c["index"], c["bits"])
            x = kern[mat + "_x"]
            y1 = h.spmv(x)
            ev, fatal = h.take_events()
            text = "".join(event_lines(ev, fmt))
            assert fatal == (c["exit"] == 1), c
            if fatal:
                assert text == c["stdout"], c
                continue
            y2 = h.spmv(x)
            ev2, _ = h.take_events()
            assert text + "".join(event_lines(ev2, fmt)) == c["stdout"], c
            assert [format(int(v), "016x") for v in y1.view(np.uint64)] == c["y1"], c
            assert [format(int(v), "016x") for v in y2.view(np.uint64)] == c["y2"], c
            assert [int(v) for v in h.ctx.stored_words(h.A)[c["index"]]] == c["words_after"][0], c
        finally:
            h.close()


@pytest.mark.parametrize("fmt", FMTS)
def test_inject_bitflip_draws_like_reference(amd, fmt, capfd):
    """-x: same libc rand() sequence as inject_bitflip (1 + num_flips draws)."""
    libc = ctypes.CDLL(None)
    cols, rows, vals, n = random_spd(64, 6, seed=12)
    for kind, name in ((0, "ANY"), (1, "VALUE"), (2, "INDEX")):
        for flips in (1, 2, 3):
            libc.srand(4321)
            o = OracleMatrix(fmt, "none", cols, rows, vals, n)
            idx, bits = o.inject_rand(kind, flips)
            ctx = amd.HIPContext("none", FNAME[fmt])
            try:
                A = ctx.create_matrix(cols, rows, vals, n, len(vals))
                libc.srand(4321)
                capfd.readouterr()
                got = ctx.inject_bitflip(A, name, flips)
                out = capfd.readouterr().out
                assert got == (idx, bits)
                assert out == "".join("*** flipping bit %d at index %d ***\n" % (b, idx) for b in bits)
                assert np.array_equal(ctx.stored_words(A), o.stored_words())
            finally:
                ctx.close()


def test_fatal_event_prints_reference_line_and_exits(amd, capfd):
    cols, rows, vals, n = laplace5(16, 16)
    ctx = amd.HIPContext("sed", "csr")
    try:
        A = ctx.create_matrix(cols, rows, vals, n, len(vals))
        vb, vx, vr, vp, vw = (ctx.create_vector(n) for _ in range(5))
        ctx.upload(vb, rhs(n, 1))
        ctx.upload(vx, np.zeros(n))
        ctx.inject_at(A, 123, [70])
        with pytest.raises(SystemExit) as e:
            amd.cg_solve(ctx, A, vb, vx, vr, vp, vw)
        assert e.value.code == 1
        assert capfd.readouterr().out == "[ECC] error detected at index 123\n"
    finally:
        ctx.close()


@pytest.mark.parametrize("fmt", FMTS)
def test_fused_dot_is_transparent(amd, fmt):
    """spmv(A,p,w) on a square matrix also forms p.w, so the dot(p,w) that follows
    costs no kernel (cross-call fusion behind the unchanged API).  The value must
    be a correct dot product whenever it is served, and must never be served
    after either vector changed."""
    cols, rows, vals, n = MATS["ragged"]()
    rng = np.random.default_rng(5)
    p, q = rng.standard_normal(n), rng.standard_normal(n)
    ctx = amd.HIPContext("secded", FNAME[fmt])
    try:
        A = ctx.create_matrix(cols, rows, vals, n, len(vals))
        vp, vw, vq = (ctx.create_vector(n) for _ in range(3))
        ctx.upload(vp, p)
        ctx.upload(vq, q)
        ctx.spmv(A, vp, vw)
        fused = ctx.dot(vp, vw)  # served by the SpMV
        w = ctx.download(vw)
        scale = float(np.abs(p * w).sum())
        assert abs(fused - ora_dot(p, w)) <= 1e-13 * scale
        assert ctx.dot(vw, vp) == fused  # either order, repeatable
        other = ctx.dot(vq, vq)  # another reduction reuses the result slot
        assert abs(other - ora_dot(q, q)) <= 1e-13 * float((q * q).sum())
        assert ctx.dot(vp, vw) == fused  # still the cached value: nothing changed
        ctx.calc_p(vp, vq, 0.5)  # p changes -> the cached product is stale
        p2 = ctx.download(vp)
        d2 = ctx.dot(vp, vw)
        assert abs(d2 - ora_dot(p2, w)) <= 1e-13 * float(np.abs(p2 * w).sum())
        assert d2 != fused
        ctx.spmv(A, vp, vw)
        ctx.upload(vw, w)  # w overwritten between spmv and dot -> must not be served
        d3 = ctx.dot(vp, vw)
        assert abs(d3 - ora_dot(p2, w)) <= 1e-13 * float(np.abs(p2 * w).sum())
        # a fused SpMV that raises an ECC event still reports it with the dot
        seen = []
        ctx.on_event = lambda ev, fatal: seen.extend(ev)
        ctx.inject_at(A, 3, [5])
        ctx.spmv(A, vp, vw)
        ctx.dot(vp, vw)
        assert seen == [(2, 3, 5)]
    finally:
        ctx.close()


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("width,rpt,lag", [(16, 8, 2), (16, 16, 0), (257, 2, 1), (100000, 4, 3), (64, 8, 2)])
def test_sweep_layout_forced_on_small_matrices(amd, fmt, width, rpt, lag, monkeypatch):
    """The sweep layout (CSR: one persistent launch over (row group, panel) segments, running
    sums in registers, paced per XCD) is normally chosen only for large scattered matrices
    (full size: test_gpu_fullsize); force it here with tiny panels, every group size and
    several pacing lags through the same checks as the streaming layout: stored elements in
    the caller's order, SpMV bit-identical, flips reported with the caller's index and
    repaired, fused dot.  COO, and CSR matrices with more than 255 elements of a row in one
    panel, take the panel layout instead (same checks)."""
    monkeypatch.setenv("ABFT_HIP_LAYOUT", "sweep")
    monkeypatch.setenv("ABFT_HIP_PANEL_WIDTH", str(width))
    monkeypatch.setenv("ABFT_HIP_SWEEP_RPT", str(rpt))
    monkeypatch.setenv("ABFT_HIP_SWEEP_LAG", str(lag))
    for mat in ("ragged", "rnd300", "lap40", "one", "tail_empty"):
        cols, rows, vals, n = MATS[mat]()
        x = rhs(n, 11) - 0.5
        for mode in ("none", "sec8", "secded"):
            o = OracleMatrix(fmt, mode, cols, rows, vals, n)
            h = Hip(amd, fmt, mode, cols, rows, vals, n)
            try:
                layout = h.ctx.matrix_info(h.A)[0]
                assert layout == ("sweep" if fmt == CSR and (mat != "ragged" or width <= 64) else "panels"), (mat, layout)
                assert np.array_equal(h.ctx.stored_words(h.A), o.stored_words())
                assert bits_equal(h.spmv(x), o.spmv(x))
                assert h.take_events() == ([], False)
                if mode != "none" and len(vals) > 2:
                    idx = [0, len(vals) // 2, len(vals) - 1]
                    for k, i in enumerate(idx):
                        o.inject(i, [7 + 30 * k])
                        h.ctx.inject_at(h.A, i, [7 + 30 * k])
                    assert bits_equal(h.spmv(x), o.spmv(x))
                    assert h.take_events() == o.events()
                    assert np.array_equal(h.ctx.stored_words(h.A), o.stored_words())
                for _ in range(2):  # again (pacing counters were reset by the last workgroup), with the fused dot
                    h.ctx.spmv(h.A, h.vx, h.vy)
                    d = h.ctx.dot(h.vx, h.vy)
                    y = h.ctx.download(h.vy)
                    assert bits_equal(y, o.spmv(x))
                    assert abs(d - ora_dot(x, y)) <= 1e-13 * float(np.abs(x * y).sum()) + 1e-300
            finally:
                h.close()


@pytest.mark.parametrize("width,srows,lag", [(16, 64, 2), (16, 16, 0), (257, 128, 1), (100000, 1024, 3), (64, 32, 2)])
def test_slice_layout_forced_on_small_matrices(amd, width, srows, lag, monkeypatch):
    """The slice layout (CSR: a wave owns a slice of rows, keeps their running sums in its own piece of
    LDS and streams the slice's elements, ordered by (panel, row), 64 per instruction; the lanes of a
    chunk that share a row are folded in lane order over DPP) is normally chosen only for large
    scattered matrices (full size: test_gpu_fullsize); force it here with tiny panels and every slice
    size through the same checks as the streaming layout: stored elements in the caller's order, SpMV
    bit-identical (rows longer than a chunk, empty rows, rows that continue in the next chunk), flips
    reported with the caller's index and repaired, fused dot, panel ranges."""
    monkeypatch.setenv("ABFT_HIP_LAYOUT", "slice")
    monkeypatch.setenv("ABFT_HIP_PANEL_WIDTH", str(width))
    monkeypatch.setenv("ABFT_HIP_SLICE_ROWS", str(srows))
    monkeypatch.setenv("ABFT_HIP_SLICE_LAG", str(lag))
    for mat in ("ragged", "rnd300", "lap40", "one", "tail_empty", "empty"):
        cols, rows, vals, n = MATS[mat]()
        x = rhs(n, 11) - 0.5
        for mode in ("none", "sed", "sec7", "sec8", "secded"):
            o = OracleMatrix(CSR, mode, cols, rows, vals, n)
            h = Hip(amd, CSR, mode, cols, rows, vals, n)
            try:
                layout = h.ctx.matrix_info(h.A)[0]
                assert layout == ("slice" if len(vals) else "stream"), (mat, layout)
                assert np.array_equal(h.ctx.stored_words(h.A), o.stored_words())
                assert bits_equal(h.spmv(x), o.spmv(x))
                assert h.take_events() == ([], False)
                if mode not in ("none", "sed") and len(vals) > 2:
                    idx = [0, len(vals) // 2, len(vals) - 1]
                    for k, i in enumerate(idx):
                        o.inject(i, [7 + 30 * k])
                        h.ctx.inject_at(h.A, i, [7 + 30 * k])
                    assert bits_equal(h.spmv(x), o.spmv(x))
                    assert h.take_events() == o.events()
                    assert np.array_equal(h.ctx.stored_words(h.A), o.stored_words())
                for _ in range(2):  # again, with the fused dot
                    h.ctx.spmv(h.A, h.vx, h.vy)
                    d = h.ctx.dot(h.vx, h.vy)
                    y = h.ctx.download(h.vy)
                    assert bits_equal(y, o.spmv(x))
                    assert abs(d - ora_dot(x, y)) <= 1e-13 * float(np.abs(x * y).sum()) + 1e-300
            finally:
                h.close()


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("width,chunk", [(16, 0), (16, 3), (257, 1), (100000, 2)])
def test_panel_layout_forced_on_small_matrices(amd, fmt, width, chunk, monkeypatch):
    """The panel (column-blocked) layout is normally chosen only for large
    scattered matrices (exercised at full size in test_gpu_fullsize); force it
    here, with tiny panels and several launch chunkings, through the same checks:
    stored elements come back in the caller's order, SpMV is bit-identical, every
    single-bit flip is reported with the caller's element index and repaired."""
    monkeypatch.setenv("ABFT_HIP_LAYOUT", "panels")
    monkeypatch.setenv("ABFT_HIP_PANEL_WIDTH", str(width))
    monkeypatch.setenv("ABFT_HIP_PANEL_CHUNK", str(chunk))
    for mat in ("ragged", "rnd300", "lap40"):
        cols, rows, vals, n = MATS[mat]()
        x = rhs(n, 11) - 0.5
        for mode in ("none", "secded"):
            o = OracleMatrix(fmt, mode, cols, rows, vals, n)
            h = Hip(amd, fmt, mode, cols, rows, vals, n)
            try:
                assert np.array_equal(h.ctx.stored_words(h.A), o.stored_words())
                assert bits_equal(h.spmv(x), o.spmv(x))
                assert h.take_events() == ([], False)
                if mode == "secded":
                    idx = [0, len(vals) // 2, len(vals) - 1]
                    for k, i in enumerate(idx):
                        o.inject(i, [7 + 30 * k])
                        h.ctx.inject_at(h.A, i, [7 + 30 * k])
                    assert bits_equal(h.spmv(x), o.spmv(x))
                    assert h.take_events() == o.events()
                    assert np.array_equal(h.ctx.stored_words(h.A), o.stored_words())
                    # fused dot through the panel kernel
                    h.ctx.spmv(h.A, h.vx, h.vy)
                    d = h.ctx.dot(h.vx, h.vy)
                    y = h.ctx.download(h.vy)
                    assert abs(d - ora_dot(x, y)) <= 1e-13 * float(np.abs(x * y).sum())
            finally:
                h.close()


@pytest.mark.parametrize("layout", ["stream", "panels", "sweep"])
@pytest.mark.parametrize("mode", ["none", "constraints", "sed", "sec7", "sec8"])
def test_coo_silently_corrupted_column_scatters_like_reference(amd, mode, layout, monkeypatch):
    """cg-coo with a column field corrupted where no check sees it (none / constraints:
    any flip that keeps the order; sed / sec8: double flips; sec7: double flips
    mis-corrected): the reference adds the product to result[corrupted col]
    (COO/CPUContext.cpp:120).  y bit for bit on two passes, several elements moving into
    the same output, out of a group, into an empty group, past the end of the vector;
    the fused vec.result product agrees with y."""
    monkeypatch.setenv("ABFT_HIP_LAYOUT", layout)
    monkeypatch.setenv("ABFT_HIP_PANEL_WIDTH", "64")
    monkeypatch.setenv("ABFT_HIP_PANEL_CHUNK", "2")
    cols, rows, vals, n = MATS["ragged"]()
    nnz = len(vals)
    x = rhs(n, 21) - 0.5
    rng = np.random.default_rng(77)
    compared = moved = 0
    for trial in range(40 if mode == "constraints" else 24):
        o = OracleMatrix(COO, mode, cols, rows, vals, n)
        h = Hip(amd, COO, mode, cols, rows, vals, n)
        try:
            nel = 1 if mode == "constraints" else int(rng.integers(1, 5))
            for _ in range(nel):
                i = int(rng.integers(0, nnz))
                if mode == "constraints":
                    bits = [int(rng.integers(0, 4))]  # small moves: some keep the (row, col) order and pass
                elif mode == "none":
                    bits = [int(rng.integers(0, 12))]  # column bits: the new column stays (mostly) below n
                    if rng.random() < 0.2:
                        bits = [int(rng.integers(10, 32))]  # ... or leaves the vector
                else:
                    bits = [int(b) for b in rng.choice(10, size=2, replace=False)]  # two column bits
                o.inject(i, bits)
                h.ctx.inject_at(h.A, i, bits)
            for _ in range(2):
                y, want = h.spmv(x), o.spmv(x)
                ev, fatal = h.take_events()
                oev, ofatal = o.events()
                assert (sorted(ev), fatal) == (sorted(oev), ofatal) or (fatal and ofatal and ev[:1] == oev[:1])
                if fatal:
                    break
                assert bits_equal(y, want), (trial, mode, layout)
                assert np.array_equal(h.ctx.stored_words(h.A), o.stored_words())
                compared += 1
                moved += int(np.any((o.stored_words()[:, 0] & (0xFFFFFF if mode not in ("none", "constraints")
                                                                else 0xFFFFFFFF)) != cols))
            if not fatal:
                h.ctx.spmv(h.A, h.vx, h.vy)
                d = h.ctx.dot(h.vx, h.vy)
                yy = h.ctx.download(h.vy)
                assert bits_equal(yy, want)
                assert abs(d - ora_dot(x, yy)) <= 1e-13 * float(np.abs(x * yy).sum())
        finally:
            h.close()
    assert compared >= 6 and moved >= 5, (compared, moved)


@pytest.mark.parametrize("layout", ["sweep", "slice"])
@pytest.mark.parametrize("mode", ["none", "secded"])
def test_spmv_by_panel_ranges_equals_one_launch(amd, mode, layout, monkeypatch):
    """abft_hip_spmv_dot_range_dev: the panel sweep cut into ranges (a shard whose input vector
    arrives slot by slot) -- the ranges in ascending order equal one launch bit for bit, y and
    the fused product; the second range only reads its own part of the input vector."""
    import ctypes as C
    from abft_sparse_cg_amd import capi
    monkeypatch.setenv("ABFT_HIP_LAYOUT", layout)
    monkeypatch.setenv("ABFT_HIP_PANEL_WIDTH", "64")
    cols, rows, vals, n = MATS["rnd300"]()
    x = rhs(n, 4)
    seen = []
    ctx = amd.HIPContext(mode, "csr", on_event=lambda ev, fatal: seen.extend(ev))
    try:
        A = ctx.create_matrix(cols, rows, vals, n, len(vals))
        if mode == "secded":
            ctx.inject_at(A, 5, [17])
        L, h = ctx.L, ctx.h
        npan, width = C.c_int(0), C.c_int(0)
        capi.check(L.abft_hip_matrix_panels(A.h, C.byref(npan), C.byref(width)))
        assert (npan.value, width.value) == ((n + 63) // 64, 64)
        vx, vy, sc = ctx.create_vector(n), ctx.create_vector(n), ctx.create_vector(2)
        ctx.upload(vx, x)
        capi.check(L.abft_hip_spmv_dot_dev(h, A.h, vx.h, vy.h, 0, sc.device_ptr))
        y0, s0 = ctx.download(vy), ctx.download(sc)
        ctx._drain()
        cut = npan.value // 2
        xa = x.copy()
        xa[cut * 64:] = np.nan  # not there yet while the first range runs
        ctx.upload(vx, xa)
        ctx.upload(vy, np.full(n, np.nan))
        capi.check(L.abft_hip_spmv_dot_range_dev(h, A.h, vx.h, vy.h, 0, sc.device_ptr, 0, cut))
        ctx.upload(vx, x)
        capi.check(L.abft_hip_spmv_dot_range_dev(h, A.h, vx.h, vy.h, 0, sc.device_ptr, cut, npan.value))
        y1, s1 = ctx.download(vy), ctx.download(sc)
        assert bits_equal(y0, y1) and s0[0] == s1[0]
        assert bits_equal(y0, OracleMatrix(CSR, mode, cols, rows, vals, n).spmv(x))
    finally:
        ctx.close()
    assert seen == ([(2, 5, 17)] if mode == "secded" else [])


@pytest.mark.parametrize("mat,lo,hi", [("lap40", 40, 1280), ("ragged", 7, 650), ("lap40", 0, 1320), ("rnd300", 100, 100)])
@pytest.mark.parametrize("mode", ["none", "secded"])
def test_spmv_in_two_parts_equals_one_launch(amd, mode, mat, lo, hi):
    """abft_hip_spmv_part / abft_hip_spmv_dot_part_dev (a shard's rows that need no
    peer data run beside the exchange): INTERIOR then BOUNDARY = ALL, bit for bit --
    y, the fused product, and the ECC events (flips on both sides of the split)."""
    from abft_sparse_cg_amd import capi
    cols, rows, vals, n = MATS[mat]()
    x = rhs(n, 4)
    nnz = len(vals)
    flips = [(0, 17), (nnz // 2, 70), (nnz - 1, 3)] if mode == "secded" else []
    seen = []
    ctx = amd.HIPContext(mode, "csr", on_event=lambda ev, fatal: seen.extend(ev))
    try:
        outs = []
        for split in (False, True):
            A = ctx.create_matrix(cols, rows, vals, n, nnz)
            if split:
                ctx.set_interior(A, lo, hi)
            for i, b in flips:
                ctx.inject_at(A, i, [b])
            vx, vy, sc = ctx.create_vector(n), ctx.create_vector(n), ctx.create_vector(2)
            ctx.upload(vx, x)
            ctx.upload(vy, np.full(n, np.nan))
            L, h = ctx.L, ctx.h
            if split:
                # interior rows are written by the first call, everything else still NaN
                capi.check(L.abft_hip_spmv_dot_part_dev(h, A.h, vx.h, vy.h, 0, sc.device_ptr, capi.PART_INTERIOR))
                part = ctx.download(vy)
                capi.check(L.abft_hip_spmv_dot_part_dev(h, A.h, vx.h, vy.h, 0, sc.device_ptr, capi.PART_BOUNDARY))
            else:
                capi.check(L.abft_hip_spmv_dot_dev(h, A.h, vx.h, vy.h, 0, sc.device_ptr))
            y, s = ctx.download(vy), ctx.download(sc)
            ctx._drain()
            outs.append((y, s, sorted(seen)))
            seen.clear()
            if split:
                done = ~np.isnan(part)
                assert not done[:lo].any() and not done[hi:].any()  # never outside the declared rows
                assert bits_equal(part[done], y[done])
                # host-scalar form: same two calls through abft_hip_spmv_part
                ctx.upload(vy, np.full(n, np.nan))
                ctx.spmv(A, vx, vy, capi.PART_INTERIOR)
                ctx.spmv(A, vx, vy, capi.PART_BOUNDARY)
                assert bits_equal(ctx.download(vy), y)
            ctx.destroy_matrix(A)
        (y0, s0, e0), (y1, s1, e1) = outs
        # (s[1], the queued-event count, is lower in the split run only because the
        # download between its two calls already drained the interior's events)
        assert bits_equal(y0, y1) and bits_equal(s0[:1], s1[:1]) and e0 == e1
        assert len(e0) == len(flips)
        o = OracleMatrix(CSR, mode, cols, rows, vals, n)
        for i, b in flips:
            o.inject(i, [b])
        assert bits_equal(y0, o.spmv(x))
    finally:
        ctx.close()


def test_deferred_x_update_is_transparent(amd):
    """calc_xr leaves x += alpha p to the calc_p that follows (cross-call fusion); any
    other call applies it first.  x, r, p must be the reference's bits on every path."""
    n = 5003
    rng = np.random.default_rng(9)
    x0, r0, p0, w0, q0 = (rng.standard_normal(n) for _ in range(5))
    alpha, beta = 0.37, -1.25
    ctx = amd.HIPContext("none", "csr")
    try:
        def fresh():
            vs = [ctx.create_vector(n) for _ in range(5)]
            for v, a in zip(vs, (x0, r0, p0, w0, q0)):
                ctx.upload(v, a)
            return vs
        xr, rr_, pr = x0.copy(), r0.copy(), p0.copy()
        want_rr = ora_calc_xr(xr, rr_, pr, w0, alpha)
        p_after = pr.copy()
        ora_calc_p(p_after, rr_, beta)

        # (1) the CG order: calc_xr then calc_p -> one fused kernel does both halves
        x, r, p, w, q = fresh()
        got = ctx.calc_xr(x, r, p, w, alpha)
        ctx.calc_p(p, r, beta)
        assert abs(got - want_rr) <= 1e-13 * want_rr
        assert bits_equal(ctx.download(x), xr) and bits_equal(ctx.download(r), rr_)
        assert bits_equal(ctx.download(p), p_after)

        # (2) x read right after calc_xr: the pending half is applied before the copy out
        x, r, p, w, q = fresh()
        ctx.calc_xr(x, r, p, w, alpha)
        assert bits_equal(ctx.download(x), xr)
        ctx.calc_p(p, r, beta)  # nothing pending any more: plain calc_p
        assert bits_equal(ctx.download(p), p_after) and bits_equal(ctx.download(x), xr)

        # (3) p overwritten before any calc_p: x must have used the old p
        x, r, p, w, q = fresh()
        ctx.calc_xr(x, r, p, w, alpha)
        ctx.copy_vector(p, q)
        assert bits_equal(ctx.download(x), xr) and bits_equal(ctx.download(p), q0)

        # (4) a calc_p on ANOTHER vector comes next: not the pair, x still right
        x, r, p, w, q = fresh()
        ctx.calc_xr(x, r, p, w, alpha)
        ctx.calc_p(q, r, beta)
        qa = q0.copy()
        ora_calc_p(qa, rr_, beta)
        assert bits_equal(ctx.download(q), qa) and bits_equal(ctx.download(x), xr)
        assert bits_equal(ctx.download(p), p0)

        # (5) two calc_xr in a row, then calc_p
        x, r, p, w, q = fresh()
        ctx.calc_xr(x, r, p, w, alpha)
        ctx.calc_xr(x, r, p, w, alpha)
        ctx.calc_p(p, r, beta)
        x2, r2 = xr.copy(), rr_.copy()
        ora_calc_xr(x2, r2, pr, w0, alpha)
        p2 = pr.copy()
        ora_calc_p(p2, r2, beta)
        assert bits_equal(ctx.download(x), x2) and bits_equal(ctx.download(p), p2)

        # (6) a vector whose device address was handed out is never deferred: read it back
        # with the HIP runtime directly -- no library call in between that could apply a
        # pending update
        x, r, p, w, q = fresh()
        ptr = x.device_ptr
        assert ptr
        ctx.calc_xr(x, r, p, w, alpha)
        hip = ctypes.CDLL(None)  # the runtime libabft_hip.so is bound to (loaded RTLD_GLOBAL)
        if not hasattr(hip, "hipMemcpy"):
            hip = ctypes.CDLL("libamdhip64.so")
        hip.hipDeviceSynchronize.restype = ctypes.c_int
        hip.hipMemcpy.restype = ctypes.c_int
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        assert hip.hipDeviceSynchronize() == 0
        seen = np.empty(n)
        assert hip.hipMemcpy(seen.ctypes.data, ptr, n * 8, 2) == 0  # hipMemcpyDeviceToHost
        assert bits_equal(seen, xr)
    finally:
        ctx.close()


def test_many_contexts_leave_the_host_heap_alone(amd):
    """Round 4: a process that creates and destroys contexts by the thousand (every test and fuzz case does) saw words
    change in 913-928-byte blocks of its own heap -- the HIP runtime still counting references on a stream object that
    hipStreamDestroy had freed (profiles/r04/stream_destroy_hunt.txt).  The library now parks a
    context's idle stream in a process-wide pool instead of destroying it.  Here: 3 000 contexts, each doing a little
    work, beside a rotating set of 916-byte canaries (the size class the stream object shares); with the destroy this
    tripped a few times per 10 000 contexts, with the pool never."""
    want = np.arange(229, dtype=np.uint32) * 2654435761 % 4294967291
    canaries = [want.copy() for _ in range(24)]
    x = np.linspace(-1.0, 1.0, 300)
    for it in range(3000):
        ctx = amd.HIPContext("none", "csr", on_event=lambda e, f: None)
        try:
            a, b = ctx.create_vector(len(x)), ctx.create_vector(len(x))
            ctx.upload(a, x)
            ctx.copy_vector(b, a)
            assert abs(ctx.dot(a, b) - float(np.dot(x, x))) < 1e-9
        finally:
            ctx.close()
        canaries[it % len(canaries)] = want.copy()  # (blocks of this size are freed and handed out again all the time)
        if it % 50 == 0:
            for c in canaries:
                assert np.array_equal(c, want), it
    for c in canaries:
        assert np.array_equal(c, want)

