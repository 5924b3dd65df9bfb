"""CPU-side checks of the drop-in boundary: libabft_hip.so loads without a GPU,
exports every symbol include/abft_hip.h declares, and refuses to run without a
device instead of falling back to anything.

Each check runs in a child interpreter so that the pytest process itself never
loads the HIP runtime (the fork-based reference tests want a small parent)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "abft_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(abft_[a-z_0-9]+)\s*\(", text)))


def child(code):
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def test_header_declares_the_whole_plugin_surface():
    syms = declared_symbols()
    # one entry point per CGContext virtual (reference CGContext.h:15-36)
    for s in ["abft_hip_matrix_create_csr", "abft_hip_matrix_create_coo", "abft_hip_matrix_destroy",
              "abft_hip_vector_create", "abft_hip_vector_destroy", "abft_hip_vector_map", "abft_hip_vector_unmap",
              "abft_hip_vector_copy", "abft_hip_dot", "abft_hip_calc_xr", "abft_hip_calc_p", "abft_hip_spmv",
              "abft_hip_inject", "abft_hip_drain_events"]:
        assert s in syms


def test_library_exports_every_declared_symbol():
    out = child("""
import ctypes
from abft_sparse_cg_amd import capi
lib = ctypes.CDLL(capi.LIB_PATH)
missing = [s for s in %r if not hasattr(lib, s)]
assert not missing, missing
assert sorted(capi.SIGNATURES) == %r
print("ok")
""" % (declared_symbols(), declared_symbols()))
    assert out.strip() == "ok"


def test_no_device_means_loud_failure_not_fallback():
    out = child("""
import ctypes
import abft_sparse_cg_amd as a
from abft_sparse_cg_amd import capi
n = ctypes.c_int(0)
capi.load().abft_hip_device_count(ctypes.byref(n))
if n.value > 0:
    print("gpu present")
else:
    try:
        a.HIPContext("none", "csr")
        print("no error")
    except a.AbftError as e:
        print(e.code, "no CPU fallback" in str(e))
""")
    assert out.strip() in ("gpu present", "-5 True")


def test_event_text_is_the_reference_text():
    out = child("""
from abft_sparse_cg_amd.capi import format_event as f
for a in [(1,7,0,0),(2,7,93,0),(3,7,0,1),(4,7,0,0),(5,3,0,0),(5,3,0,1),(6,3,0,0),(6,3,0,1),(7,3,0,0),(7,3,0,1),(8,3,0,0),(8,3,0,1)]:
    print(f(*a), end="")
""")
    assert out == ("[ECC] error detected at index 7\n"
                   "[ECC] corrected bit 93 at index 7\n"
                   "[ECC] corrected overall parity bit at index 7\n"
                   "[ECC] double-bit error detected\n"
                   "row size constraint violated for row 3\n"
                   "row size constraint violated for index 3\n"
                   "row order constraint violated for row3\n"
                   "row index order violated at index 3\n"
                   "column size constraint violated at index 3\n"
                   "column size constraint violated for index 3\n"
                   "column order constraint violated at index 3\n"
                   "column index order violated at index 3\n")


def test_generators_are_self_consistent():
    import numpy as np
    sys.path.insert(0, ROOT)
    from abft_sparse_cg_amd import generators as g  # plain C++ library, no HIP
    from _oracle import laplace5
    c, r, v, n = g.generate("laplace5:9,7")
    c2, r2, v2, n2 = laplace5(9, 7)
    assert n == n2 and np.array_equal(c, c2) and np.array_equal(r, r2) and np.array_equal(v, v2)
    for spec in ("random:1024,8,1", "powerlaw:2048,3"):
        c, r, v, n = g.generate(spec)
        key = r.astype(np.int64) * n + c
        assert np.all(np.diff(key) > 0)  # sorted by (row, col), no duplicates
        tkey = c.astype(np.int64) * n + r
        order = np.argsort(tkey)
        assert np.array_equal(tkey[order], key) and np.array_equal(v[order], v)  # symmetric
        diag = v[r == c]
        off = np.bincount(r[r != c], weights=-v[r != c], minlength=n)
        assert len(diag) == n and np.all(diag - off > 0.99)  # strictly diagonally dominant
        b = g.partition(spec, 4)
        parts = [g.generate(spec, b[k], b[k + 1]) for k in range(4)]
        assert np.array_equal(np.concatenate([p[0] for p in parts]), c)
        assert np.array_equal(np.concatenate([p[2] for p in parts]), v)
        sizes = [len(p[2]) for p in parts]
        assert max(sizes) - min(sizes) <= 200
    assert g.dim("laplace5:3162,3162") == 9998244
