"""CPU baseline for bench.py's `cpu_baseline` leg (TEST/MEASUREMENT
INFRASTRUCTURE -- the product never imports this).

Times the CG loop of the reference's own CSR CPUContext (oracle/_ref/
libref_csr.so, built from /root/reference by oracle/Makefile: kind
"reference"), or, if that build is absent, of our C restatement
(oracle/libabft_oracle.so: kind "port"), on the same in-memory matrix the GPU
run uses.  Structure is the reference's: OpenMP over rows in spmv, serial
dot / calc_xr / calc_p (CSR/CPUContext.cpp:82-133).
"""
import ctypes as C
import os
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
u32p = C.POINTER(C.c_uint32)
f64p = C.POINTER(C.c_double)


def _p(a, t):
    return a.ctypes.data_as(t)


def host_cores():
    """CPU threads this job may use: the cgroup quota if there is one, else the
    affinity mask, capped at 16 (a one-GPU box's CPU share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def _reference_rhs(n):
    """glibc rand() / RAND_MAX with the default seed: the reference driver's b (cg.cpp:66-74)"""
    import sys
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from abft_sparse_cg_amd import generators
    return generators.reference_rhs(n)


def time_cg(cols, rows, vals, n, mode, iters, threads=None, runs=1):
    """-> dict(kind, iters, seconds, it_per_s, it_per_s_runs, cores).  Runs `iters` CG iterations
    (conv threshold 0) from x=0, b = deterministic rhs, `runs` times on one matrix (the reference's
    run_benchmark repeats whole program runs; its clock, like this one, covers the CG loop only)."""
    threads = threads or host_cores()
    os.environ["OMP_NUM_THREADS"] = str(threads)
    try:  # the OpenMP runtime reads the variable once; set the count directly for later calls
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
    except OSError:
        pass
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    rows = np.ascontiguousarray(rows, dtype=np.uint32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    b = _reference_rhs(n)
    x = np.zeros(n)
    r, p, w = np.empty(n), np.empty(n), np.empty(n)
    hist = np.zeros(max(iters, 1))
    ref_so = os.path.join(HERE, "_ref", "libref_csr.so")
    if os.path.exists(ref_so):
        L = C.CDLL(ref_so)
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.c_char_p]
        L.ref_matrix_create.restype = C.c_void_p
        L.ref_matrix_create.argtypes = [C.c_void_p, u32p, u32p, f64p, C.c_int, C.c_int]
        L.ref_cg.restype = C.c_int
        L.ref_cg.argtypes = [C.c_void_p, C.c_void_p, f64p, f64p, f64p, f64p, f64p, C.c_int, C.c_int, C.c_double, f64p]
        L.ref_matrix_destroy.argtypes = [C.c_void_p, C.c_void_p]
        ctx = L.ref_create(mode.encode())
        mat = L.ref_matrix_create(ctx, _p(cols, u32p), _p(rows, u32p), _p(vals, f64p), n, len(vals))
        dts = []
        for _ in range(max(runs, 1)):
            x[:] = 0.0
            t0 = time.perf_counter()
            it = L.ref_cg(ctx, mat, _p(b, f64p), _p(x, f64p), _p(r, f64p), _p(p, f64p), _p(w, f64p), n, iters, 0.0,
                          _p(hist, f64p))
            dts.append(time.perf_counter() - t0)
        L.ref_matrix_destroy(ctx, mat)
        kind = "reference"
    else:
        so = os.path.join(HERE, "libabft_oracle.so")
        L = C.CDLL(so)
        L.ora_matrix_create.restype = C.c_void_p
        L.ora_matrix_create.argtypes = [C.c_int, C.c_int, u32p, u32p, f64p, C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.ora_cg.restype = C.c_int
        L.ora_cg.argtypes = [C.c_void_p, f64p, f64p, f64p, f64p, f64p, C.c_int, C.c_double, f64p, C.c_int,
                             C.POINTER(C.c_int)]
        L.ora_matrix_destroy.argtypes = [C.c_void_p]
        modes = ["none", "constraints", "sed", "sec7", "sec8", "secded"]
        mat = L.ora_matrix_create(0, modes.index(mode), _p(cols, u32p), _p(rows, u32p), _p(vals, f64p), n, n,
                                  len(vals), 0)
        fatal = C.c_int(0)
        dts = []
        for _ in range(max(runs, 1)):
            x[:] = 0.0
            t0 = time.perf_counter()
            it = L.ora_cg(mat, _p(b, f64p), _p(x, f64p), _p(r, f64p), _p(p, f64p), _p(w, f64p), iters, 0.0,
                          _p(hist, f64p), threads, C.byref(fatal))
            dts.append(time.perf_counter() - t0)
        L.ora_matrix_destroy(mat)
        kind = "port"
    dt = sum(dts)
    return {"kind": kind, "iters": int(it), "seconds": dt, "it_per_s": it * len(dts) / dt if dt > 0 else 0.0,
            "it_per_s_runs": [it / d if d > 0 else 0.0 for d in dts],
            "cores": int(threads), "rr_last": float(hist[it - 1]) if it else None}
