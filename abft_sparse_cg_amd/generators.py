"""Synthetic matrices of BASELINE.json's configurations (ctypes view of
host/generators.cpp, libabft_host.so): sorted symmetric COO triplets for any row
range, which is what the reference loader hands to create_matrix
(cg.cpp:394-418)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libabft_host.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `make -C %s`" % (LIB_PATH, os.path.join(_HERE, "host")))
        L = C.CDLL(LIB_PATH)
        L.abft_gen_dim.restype = C.c_int64
        L.abft_gen_dim.argtypes = [C.c_char_p]
        L.abft_gen_count.restype = C.c_int64
        L.abft_gen_count.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p]
        L.abft_gen_fill.restype = C.c_int64
        L.abft_gen_fill.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.abft_gen_partition.restype = C.c_int
        L.abft_gen_partition.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.abft_glibc_rand_fill.argtypes = [C.c_void_p, C.c_int64, C.c_uint]
        L.abft_load_mtx.restype = C.c_int
        L.abft_load_mtx.argtypes = [C.c_char_p, C.c_int] + [C.c_void_p] * 6
        L.abft_free_triplets.argtypes = [C.c_void_p] * 3
        _lib = L
    return _lib


def dim(spec):
    n = load().abft_gen_dim(spec.encode())
    if n < 0:
        raise ValueError("bad matrix spec %r" % spec)
    return int(n)


def generate(spec, row0=0, row1=None):
    """-> (cols, rows, vals, N): triplets of rows [row0,row1), sorted by (row,col),
    global indices."""
    L = load()
    n = dim(spec)
    row1 = n if row1 is None else row1
    nnz = L.abft_gen_count(spec.encode(), row0, row1, None)
    if nnz < 0:
        raise ValueError("bad row range [%d,%d) for %r" % (row0, row1, spec))
    cols = np.empty(nnz, dtype=np.uint32)
    rows = np.empty(nnz, dtype=np.uint32)
    vals = np.empty(nnz, dtype=np.float64)
    got = L.abft_gen_fill(spec.encode(), row0, row1, cols.ctypes.data, rows.ctypes.data, vals.ctypes.data)
    assert got == nnz
    return cols, rows, vals, n


def partition(spec, parts):
    """Row-block boundaries of `parts` shards with (nearly) equal nnz."""
    b = np.zeros(parts + 1, dtype=np.int64)
    if load().abft_gen_partition(spec.encode(), parts, b.ctypes.data) != 0:
        raise ValueError("bad matrix spec %r" % spec)
    return [int(v) for v in b]


def reference_rhs(n, seed=1):
    """b[i] = rand() / RAND_MAX for glibc's rand() after srand(seed): the right-hand
    side the reference driver builds (cg.cpp:66-74)."""
    out = np.empty(n, dtype=np.float64)
    load().abft_glibc_rand_fill(out.ctypes.data, n, seed)
    return out


def load_mtx(path, num_blocks=1):
    """Matrix-Market file in the reference loader's dialect (cg.cpp:342-418):
    -> (cols, rows, vals, N, block_size)."""
    L = load()
    n, bs, nnz = C.c_int(), C.c_int(), C.c_int()
    pc, pr, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = L.abft_load_mtx(path.encode(), num_blocks, C.byref(n), C.byref(bs), C.byref(nnz), C.byref(pc), C.byref(pr),
                         C.byref(pv))
    if rc == 1:
        raise FileNotFoundError(path)
    if rc == 2:
        raise ValueError("Matrix is not square")
    if rc != 0:
        raise ValueError("Failed to read matrix data")
    k = nnz.value
    cols = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint32)), shape=(max(k, 1),))[:k].copy()
    rows = np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_uint32)), shape=(max(k, 1),))[:k].copy()
    vals = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_double)), shape=(max(k, 1),))[:k].copy()
    L.abft_free_triplets(pc, pr, pv)
    return cols, rows, vals, n.value, bs.value
