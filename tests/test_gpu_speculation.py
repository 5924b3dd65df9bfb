"""Speculation behind the unchanged host-scalar API (opt-in, ABFT_HIP_SPECULATE=1; include/abft_hip.h:
abft_hip_speculation_stats; the reference loop cg.cpp:97-112): from the second iteration on the library runs the r half and the x / p half of an iteration
ahead of the caller's calc_xr / calc_p, into shadow buffers, and swaps them in when the calls arrive as predicted.
Transparent: every vector and every scalar the same bits as with ABFT_HIP_SPECULATE=0, whatever the caller does
between the calls -- the predicted continuation, another alpha or beta, a look at a vector in the middle, a stop
after calc_xr."""
import ctypes as C

import numpy as np
import pytest

from _oracle import laplace5, random_spd

pytestmark = pytest.mark.gpu


def bits(a):
    return np.asarray(a).view(np.uint64)


def stats(ctx):
    from abft_sparse_cg_amd import capi
    t, d = C.c_long(), C.c_long()
    capi.check(ctx.L.abft_hip_speculation_stats(ctx.h, C.byref(t), C.byref(d)))
    return t.value, d.value


def cg(amd, fmt, mode, mat, iters, script=None):
    """`iters` iterations of the reference loop through the C ABI; script[k] perturbs iteration k:
    'alpha' / 'beta' (one ulp off: not the quotient), 'peek' (download x between calc_xr and calc_p),
    'nodot' (the caller forms p.w with a dot of its own vectors' copies -- i.e. does not call dot(p, w)), 'stop'."""
    from abft_sparse_cg_amd.context import fdiv
    cols, rows, vals, n = mat
    ctx = amd.HIPContext(mode, fmt)
    A = ctx.create_matrix(cols, rows, vals, n, len(vals))
    b, x, r, p, w = (ctx.create_vector(n) for _ in range(5))
    ctx.upload(b, np.random.default_rng(4).random(n))
    ctx.upload(x, np.zeros(n))
    ctx.copy_vector(r, b)
    ctx.copy_vector(p, r)
    rr = ctx.dot(r, r)
    hist, peeks = [rr], []
    for k in range(iters):
        what = (script or {}).get(k, "")
        ctx.spmv(A, p, w)
        pw = ctx.dot(p, w)
        alpha = fdiv(rr, pw)
        if what == "alpha":
            alpha = float(np.nextafter(alpha, 2 * alpha))
        rr_new = ctx.calc_xr(x, r, p, w, alpha)
        if what == "peek":
            peeks.append(ctx.download(x))
        if what == "stop":
            hist.append(rr_new)
            break
        beta = fdiv(rr_new, rr)
        if what == "beta":
            beta = float(np.nextafter(beta, 2 * beta))
        ctx.calc_p(p, r, beta)
        rr = rr_new
        hist.append(rr)
    out = [ctx.download(v) for v in (x, r, p, w)]
    st = stats(ctx)
    ctx.close()
    return hist, out, peeks, st


CASES = [("csr", "secded", "lap"), ("csr", "none", "rand"), ("coo", "sec7", "lap")]


@pytest.mark.parametrize("fmt,mode,which", CASES)
def test_speculated_iterations_leave_the_same_bits(fmt, mode, which, monkeypatch):
    import abft_sparse_cg_amd as amd
    mat = laplace5(70, 53) if which == "lap" else random_spd(20011, 9, seed=3)
    scripts = [None, {3: "alpha"}, {4: "beta"}, {2: "peek", 5: "peek"}, {6: "stop"}, {1: "alpha", 2: "beta", 3: "peek", 7: "stop"}]
    for script in scripts:
        monkeypatch.setenv("ABFT_HIP_SPECULATE", "0")
        h0, v0, p0, s0 = cg(amd, fmt, mode, mat, 9, script)
        monkeypatch.setenv("ABFT_HIP_SPECULATE", "1")
        h1, v1, p1, s1 = cg(amd, fmt, mode, mat, 9, script)
        assert s0 == (0, 0)
        assert np.array_equal(bits(h0), bits(h1)), script
        for a, b in zip(v0 + p0, v1 + p1):
            assert np.array_equal(bits(a), bits(b)), script
        if script is None:
            assert s1 == (8, 0)  # every iteration from the second on was run ahead and taken over
        else:
            assert s1[0] >= 3 and s1[1] >= 1, (script, s1)  # the perturbed ones were dropped, the others taken over


def test_speculation_keeps_out_of_the_way(monkeypatch):
    """vectors the caller can see into (a raw device pointer handed out, a view) are never speculated on; destroying a
    vector forgets the learned iteration; the device-scalar loop and graph capture do not speculate"""
    import abft_sparse_cg_amd as amd
    monkeypatch.setenv("ABFT_HIP_SPECULATE", "1")
    mat = laplace5(40, 40)
    cols, rows, vals, n = mat
    ctx = amd.HIPContext("secded", "csr")
    A = ctx.create_matrix(cols, rows, vals, n, len(vals))
    b, x, r, p, w = (ctx.create_vector(n) for _ in range(5))
    ctx.upload(b, np.random.default_rng(1).random(n))
    ctx.upload(x, np.zeros(n))
    ctx.copy_vector(r, b)
    ctx.copy_vector(p, r)
    _ = r.device_ptr  # exposed
    it, rr = amd.cg_solve(ctx, A, b, x, r, p, w, max_itrs=12, conv_threshold=0.0)
    assert it == 12 and stats(ctx) == (0, 0)
    ctx.close()
