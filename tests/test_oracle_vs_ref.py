"""Pins oracle/abft_oracle.c against the REFERENCE's own CPUContext objects
(oracle/_ref/, built from /root/reference by oracle/Makefile).  Skipped when
that build is absent; tests/test_golden.py then pins the oracle through the
committed fixtures the same build produced."""
import numpy as np
import pytest

from _capture import run_captured
from _oracle import (COO, CSR, MODES, Oracle, OracleMatrix, Ref, event_lines, have_ref, laplace5,
                     ora_calc_p, ora_calc_xr, ora_dot, random_spd, ref_cg, ref_flip_spmv, ref_inject_rand, rhs)

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")]

FMTS = [CSR, COO]
NBITS = {CSR: 96, COO: 128}


def _rand_words(fmt, rng):
    n = 3 if fmt == CSR else 4
    w = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    ew = 2 if fmt == CSR else 0
    w[ew] &= 0x00FFFFFF
    return w


@pytest.mark.parametrize("fmt", FMTS)
def test_syndrome_parity_match_reference_on_random_words(fmt):
    rng = np.random.default_rng(7)
    L = Ref.lib(fmt)
    for _ in range(2000):
        n = 3 if fmt == CSR else 4
        w = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
        p = w.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_uint32))
        assert Oracle.syndrome(fmt, w) == L.ref_ecc_syndrome(p)
        assert Oracle.parity(fmt, w) == L.ref_ecc_parity(p)


@pytest.mark.parametrize("fmt", FMTS)
def test_flipped_bit_matches_reference_for_every_syndrome(fmt):
    L = Ref.lib(fmt)
    for h in range(1, 128):
        s = 0
        for p in range(1, 8):
            if (h >> (p - 1)) & 1:
                s |= 1 << (32 - p)
        assert Oracle.flipped_bit(fmt, s) == L.ref_ecc_flipped_bit(s), h


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("mode", MODES)
def test_encoded_matrix_identical(fmt, mode):
    cols, rows, vals, n = random_spd(60, 6, seed=3)
    o = OracleMatrix(fmt, mode, cols, rows, vals, n)
    r = Ref(fmt, mode, cols, rows, vals, n)
    assert np.array_equal(o.stored_words(), r.stored_words())
    if fmt == CSR:
        assert np.array_equal(o.csr_arrays()[1], r.rowptr())


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("mode", MODES)
def test_spmv_bit_identical_no_faults(fmt, mode):
    for mat in (laplace5(9, 7), random_spd(80, 8, seed=5)):
        cols, rows, vals, n = mat
        x = rhs(n, 11) - 0.5
        o = OracleMatrix(fmt, mode, cols, rows, vals, n)
        code, text, ((y_ref,), _) = run_captured(ref_flip_spmv, fmt, mode, mat, 0, [], x, 1)
        assert code == 0 and text == ""
        y = o.spmv(x)
        assert np.array_equal(y.view(np.uint64), y_ref.view(np.uint64))
        assert o.events() == ([], False)


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("mode", ["sed", "sec7", "sec8", "secded"])
def test_every_single_flip_same_events_and_result(fmt, mode):
    mat = random_spd(24, 4, seed=9)
    cols, rows, vals, n = mat
    x = rhs(n, 2) + 0.25
    index = len(vals) // 2
    for bit in range(NBITS[fmt]):
        code, text, res = run_captured(ref_flip_spmv, fmt, mode, mat, index, [bit], x)
        o = OracleMatrix(fmt, mode, cols, rows, vals, n)
        o.inject(index, [bit])
        y1 = o.spmv(x)
        ev, fatal = o.events()
        lines = event_lines(ev, fmt)
        if fatal:
            assert code == 1, (bit, text)
            assert text == "".join(lines)
        else:
            assert code == 0, (bit, text)
            y2 = o.spmv(x)
            ev2, _ = o.events()
            assert text == "".join(lines + event_lines(ev2, fmt)), bit
            (r1, r2), words = res
            assert np.array_equal(y1.view(np.uint64), r1.view(np.uint64)), bit
            assert np.array_equal(y2.view(np.uint64), r2.view(np.uint64)), bit
            assert np.array_equal(o.stored_words(), words), bit


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("mode", ["sec8", "secded"])
def test_double_flips_same_behaviour(fmt, mode):
    mat = random_spd(24, 4, seed=10)
    cols, rows, vals, n = mat
    x = rhs(n, 3) + 0.25
    index = 7
    rng = np.random.default_rng(1)
    for _ in range(60):
        b1, b2 = rng.choice(NBITS[fmt], size=2, replace=False)
        # a double flip that changes the gather index beyond the vector is UB in the reference
        code, text, res = run_captured(ref_flip_spmv, fmt, mode, mat, index, [b1, b2], x, 1)
        o = OracleMatrix(fmt, mode, cols, rows, vals, n)
        o.inject(index, [b1, b2])
        y = o.spmv(x)
        ev, fatal = o.events()
        if mode == "secded":
            assert fatal and code == 1
            assert text == "[ECC] double-bit error detected\n" == "".join(event_lines(ev, fmt))
        else:
            assert not fatal and ev == []  # sec8 is blind to double flips (parity 0)
            if code == 0:
                assert text == ""


@pytest.mark.parametrize("fmt", FMTS)
def test_constraints_violations_match(fmt):
    mat = random_spd(30, 6, seed=4)
    cols, rows, vals, n = mat
    x = rhs(n, 5)
    idx_bits = range(64, 96) if fmt == CSR else range(0, 64)
    hit = 0
    for index in (0, 5, len(vals) - 1, len(vals) // 3):
        for bit in idx_bits:
            code, text, res = run_captured(ref_flip_spmv, fmt, "constraints", mat, index, [bit], x, 1)
            if code not in (0, 1):
                continue  # reference faulted on an out-of-range gather before any check fired
            o = OracleMatrix(fmt, "constraints", cols, rows, vals, n)
            o.inject(index, [bit])
            y = o.spmv(x)
            ev, fatal = o.events()
            assert (code == 1) == fatal, (index, bit, text)
            assert text == "".join(event_lines(ev, fmt)), (index, bit)
            if not fatal:
                assert np.array_equal(y.view(np.uint64), res[0][0].view(np.uint64))
            hit += fatal
    assert hit > 20


def test_vector_kernels_bit_identical():
    cols, rows, vals, n = laplace5(5, 5)
    r = Ref(CSR, "none", cols, rows, vals, n)
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal(1000), rng.standard_normal(1000)
    assert ora_dot(a, b) == r.dot(a, b)
    x1, r1, p1, w1 = (rng.standard_normal(1000) for _ in range(4))
    x2, r2 = x1.copy(), r1.copy()
    assert ora_calc_xr(x1, r1, p1, w1, 0.37) == r.calc_xr(x2, r2, p1, w1, 0.37)
    assert np.array_equal(x1, x2) and np.array_equal(r1, r2)
    p2 = p1.copy()
    ora_calc_p(p1, r1, 1.7)
    r.calc_p(p2, r2, 1.7)
    assert np.array_equal(p1, p2)


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("mode", MODES)
def test_cg_history_bit_identical(fmt, mode):
    mat = laplace5(16, 16)
    cols, rows, vals, n = mat
    b = rhs(n, 1)
    o = OracleMatrix(fmt, mode, cols, rows, vals, n)
    it_o, h_o, x_o, fatal = o.cg(b)
    code, text, (it_r, h_r, x_r) = run_captured(ref_cg, fmt, mode, mat, b)
    assert code == 0 and not fatal
    assert it_o == it_r and it_o > 10
    assert np.array_equal(h_o.view(np.uint64), h_r.view(np.uint64))
    assert np.array_equal(x_o.view(np.uint64), x_r.view(np.uint64))


@pytest.mark.parametrize("fmt", FMTS)
def test_inject_rand_draws_like_reference(fmt):
    """Same libc rand() sequence -> same element and bits as inject_bitflip."""
    import ctypes
    libc = ctypes.CDLL(None)
    mat = random_spd(30, 6, seed=12)
    cols, rows, vals, n = mat

    for kind in (0, 1, 2):
        for flips in (1, 2, 3):
            code, text, words = run_captured(ref_inject_rand, fmt, mat, 1234, kind, flips)
            assert code == 0
            libc.srand(1234)
            o = OracleMatrix(fmt, "none", cols, rows, vals, n)
            idx, bits = o.inject_rand(kind, flips)
            assert text == "".join("*** flipping bit %d at index %d ***\n" % (b, idx) for b in bits)
            assert np.array_equal(o.stored_words(), words)
