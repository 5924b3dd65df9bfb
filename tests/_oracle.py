"""ctypes bindings for the CPU checkers (test infrastructure only).

`Oracle`  -> oracle/libabft_oracle.so   (our C restatement, oracle/abft_oracle.c)
`Ref`     -> oracle/_ref/libref_{csr,coo}.so (the reference's own CPUContext,
             built by oracle/Makefile from /root/reference; may be absent)
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

MODES = ["none", "constraints", "sed", "sec7", "sec8", "secded"]
MODE_ID = {m: i for i, m in enumerate(MODES)}
CSR, COO = 0, 1
FMT_NAME = {CSR: "csr", COO: "coo"}

EV_SED, EV_BIT, EV_PARITY, EV_DOUBLE, EV_ROW_SIZE, EV_ROW_ORDER, EV_COL_SIZE, EV_COL_ORDER = range(1, 9)

u32p = C.POINTER(C.c_uint32)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int)


class Event(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32), ("bit", C.c_uint32), ("fmt", C.c_uint32)]

    def tup(self):
        return (self.kind, self.index, self.bit)


def _p(a, t):
    return a.ctypes.data_as(t)


def build_oracle():
    so = os.path.join(ORACLE_DIR, "libabft_oracle.so")
    src = os.path.join(ORACLE_DIR, "abft_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libabft_oracle.so"], stdout=subprocess.DEVNULL)
    return so


class Oracle:
    """Thin object wrapper over oracle/abft_oracle.h."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(build_oracle())
            L.ora_ecc_masks.argtypes = [C.c_int, u32p]
            L.ora_ecc_syndrome.argtypes = [C.c_int, u32p]
            L.ora_ecc_syndrome.restype = C.c_uint32
            L.ora_ecc_parity.argtypes = [C.c_int, u32p]
            L.ora_ecc_parity.restype = C.c_uint32
            L.ora_ecc_flipped_bit.argtypes = [C.c_int, C.c_uint32]
            L.ora_ecc_flipped_bit.restype = C.c_uint32
            L.ora_ecc_encode.argtypes = [C.c_int, C.c_int, u32p]
            L.ora_csr_encode_col.argtypes = [C.c_int, C.c_uint64, C.c_uint32]
            L.ora_csr_encode_col.restype = C.c_uint32
            L.ora_coo_encode_col.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64]
            L.ora_coo_encode_col.restype = C.c_uint32
            L.ora_matrix_create.argtypes = [C.c_int, C.c_int, u32p, u32p, f64p, C.c_int, C.c_int, C.c_int, C.c_uint32]
            L.ora_matrix_create.restype = C.c_void_p
            L.ora_matrix_destroy.argtypes = [C.c_void_p]
            for n in ("ora_matrix_csr_cols", "ora_matrix_csr_rowptr", "ora_matrix_csr_values", "ora_matrix_coo_elements"):
                getattr(L, n).argtypes = [C.c_void_p]
                getattr(L, n).restype = C.c_void_p
            L.ora_inject.argtypes = [C.c_void_p, C.c_uint32, i32p, C.c_int]
            L.ora_inject_rand.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p]
            L.ora_inject_rand.restype = C.c_int
            L.ora_spmv.argtypes = [C.c_void_p, f64p, f64p, C.c_int]
            L.ora_spmv.restype = C.c_int
            L.ora_events.argtypes = [C.c_void_p, C.POINTER(Event), C.c_int, i32p]
            L.ora_events.restype = C.c_int
            L.ora_dot.argtypes = [f64p, f64p, C.c_int]
            L.ora_dot.restype = C.c_double
            L.ora_calc_xr.argtypes = [f64p, f64p, f64p, f64p, C.c_double, C.c_int]
            L.ora_calc_xr.restype = C.c_double
            L.ora_calc_p.argtypes = [f64p, f64p, C.c_double, C.c_int]
            L.ora_cg.argtypes = [C.c_void_p, f64p, f64p, f64p, f64p, f64p, C.c_int, C.c_double, f64p, C.c_int, i32p]
            L.ora_cg.restype = C.c_int
            L.ora_format_event.argtypes = [C.POINTER(Event), C.c_char_p, C.c_size_t]
            L.ora_event_is_fatal.argtypes = [C.c_uint32]
            cls._lib = L
        return cls._lib

    # ---- bit level ----
    @classmethod
    def masks(cls, fmt):
        out = np.zeros((7, 4), dtype=np.uint32)
        cls.lib().ora_ecc_masks(fmt, _p(out, u32p))
        return out

    @classmethod
    def syndrome(cls, fmt, words):
        w = np.ascontiguousarray(words, dtype=np.uint32)
        return cls.lib().ora_ecc_syndrome(fmt, _p(w, u32p))

    @classmethod
    def parity(cls, fmt, words):
        w = np.ascontiguousarray(words, dtype=np.uint32)
        return cls.lib().ora_ecc_parity(fmt, _p(w, u32p))

    @classmethod
    def flipped_bit(cls, fmt, syndrome):
        return cls.lib().ora_ecc_flipped_bit(fmt, syndrome)

    @classmethod
    def encode(cls, fmt, mode, words):
        w = np.array(words, dtype=np.uint32)
        cls.lib().ora_ecc_encode(fmt, MODE_ID[mode] if isinstance(mode, str) else mode, _p(w, u32p))
        return w

    @classmethod
    def format_event(cls, ev):
        buf = C.create_string_buffer(128)
        cls.lib().ora_format_event(C.byref(ev), buf, 128)
        return buf.value.decode()


def event_lines(events, fmt):
    out = []
    for k, i, b in events:
        e = Event(k, i, b, fmt)
        out.append(Oracle.format_event(e))
    return out


class OracleMatrix:
    def __init__(self, fmt, mode, cols, rows, vals, n_out, n_in=None, index_base=0):
        L = Oracle.lib()
        self.L = L
        self.fmt, self.mode = fmt, mode
        self.n_out = int(n_out)
        self.n_in = int(n_in if n_in is not None else n_out)
        self.nnz = int(len(vals))
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        self.h = L.ora_matrix_create(fmt, MODE_ID[mode], _p(cols, u32p), _p(rows, u32p), _p(vals, f64p),
                                     self.n_out, self.n_in, self.nnz, index_base)

    def close(self):
        if self.h:
            self.L.ora_matrix_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _view(self, fn, dtype, n):
        ptr = getattr(self.L, fn)(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,))

    def csr_arrays(self):
        return (self._view("ora_matrix_csr_cols", np.uint32, self.nnz).copy(),
                self._view("ora_matrix_csr_rowptr", np.uint32, self.n_out + 1).copy(),
                self._view("ora_matrix_csr_values", np.float64, self.nnz).copy())

    def coo_words(self):
        return self._view("ora_matrix_coo_elements", np.uint32, 4 * self.nnz).copy().reshape(-1, 4)

    def stored_words(self):
        """(nnz, 3|4) uint32 image of every stored element, reference word order."""
        if self.fmt == CSR:
            c, _, v = self.csr_arrays()
            w = np.empty((self.nnz, 3), dtype=np.uint32)
            w[:, :2] = v.view(np.uint32).reshape(-1, 2)
            w[:, 2] = c
            return w
        return self.coo_words()

    def inject(self, index, bits):
        b = np.ascontiguousarray(bits, dtype=np.int32)
        self.L.ora_inject(self.h, index, _p(b, i32p), len(b))

    def inject_rand(self, kind, num_flips):
        b = np.zeros(max(1, num_flips), dtype=np.int32)
        idx = self.L.ora_inject_rand(self.h, kind, num_flips, _p(b, i32p))
        return idx, list(b[:num_flips])

    def spmv(self, x, threads=1):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert len(x) == self.n_in
        y = np.full(self.n_out, np.nan)
        self.L.ora_spmv(self.h, _p(x, f64p), _p(y, f64p), threads)
        return y

    def events(self):
        buf = (Event * 4096)()
        fatal = C.c_int(0)
        n = self.L.ora_events(self.h, buf, 4096, C.byref(fatal))
        return [buf[i].tup() for i in range(n)], bool(fatal.value)

    def cg(self, b, max_itrs=1000, conv=1e-3, threads=1):
        n = self.n_out
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros(n)
        r = np.empty(n)
        p = np.empty(n)
        w = np.empty(n)
        hist = np.zeros(max(1, max_itrs))
        fatal = C.c_int(0)
        it = self.L.ora_cg(self.h, _p(b, f64p), _p(x, f64p), _p(r, f64p), _p(p, f64p), _p(w, f64p),
                           max_itrs, conv, _p(hist, f64p), threads, C.byref(fatal))
        return it, hist[:it].copy(), x, bool(fatal.value)


def ora_dot(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return Oracle.lib().ora_dot(_p(a, f64p), _p(b, f64p), len(a))


def ora_calc_xr(x, r, p, w, alpha):
    """in place on x, r (float64 contiguous); returns r.r"""
    return Oracle.lib().ora_calc_xr(_p(x, f64p), _p(r, f64p), _p(p, f64p), _p(w, f64p), alpha, len(x))


def ora_calc_p(p, r, beta):
    Oracle.lib().ora_calc_p(_p(p, f64p), _p(r, f64p), beta, len(p))


# ---------------------------------------------------------------- reference --

def ref_path(fmt):
    return os.path.join(ORACLE_DIR, "_ref", "libref_%s.so" % FMT_NAME[fmt])


def ref_exe(fmt):
    return os.path.join(ORACLE_DIR, "_ref", "cg-%s-ref" % FMT_NAME[fmt])


def have_ref():
    return all(os.path.exists(ref_path(f)) for f in (CSR, COO))


class Ref:
    """The reference's CPUContext (one (format, mode)) behind oracle/ref_harness.cpp.

    Its printf output goes to the process's fd 1 and fatal events exit(1): use
    tests/_capture.run_captured for calls that may print or die."""

    _libs = {}

    @classmethod
    def lib(cls, fmt):
        if fmt not in cls._libs:
            L = C.CDLL(ref_path(fmt))
            L.ref_create.argtypes = [C.c_char_p]
            L.ref_create.restype = C.c_void_p
            L.ref_destroy.argtypes = [C.c_void_p]
            L.ref_matrix_create.argtypes = [C.c_void_p, u32p, u32p, f64p, C.c_int, C.c_int]
            L.ref_matrix_create.restype = C.c_void_p
            L.ref_matrix_destroy.argtypes = [C.c_void_p, C.c_void_p]
            L.ref_matrix_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            L.ref_flip.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
            L.ref_inject_rand.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            L.ref_spmv.argtypes = [C.c_void_p, C.c_void_p, f64p, f64p, C.c_int]
            L.ref_dot.argtypes = [C.c_void_p, f64p, f64p, C.c_int]
            L.ref_dot.restype = C.c_double
            L.ref_calc_xr.argtypes = [C.c_void_p, f64p, f64p, f64p, f64p, C.c_double, C.c_int]
            L.ref_calc_xr.restype = C.c_double
            L.ref_calc_p.argtypes = [C.c_void_p, f64p, f64p, C.c_double, C.c_int]
            L.ref_cg.argtypes = [C.c_void_p, C.c_void_p, f64p, f64p, f64p, f64p, f64p, C.c_int, C.c_int, C.c_double, f64p]
            L.ref_cg.restype = C.c_int
            L.ref_ecc_syndrome.argtypes = [u32p]
            L.ref_ecc_syndrome.restype = C.c_uint32
            L.ref_ecc_parity.argtypes = [u32p]
            L.ref_ecc_parity.restype = C.c_uint32
            L.ref_ecc_flipped_bit.argtypes = [C.c_uint32]
            L.ref_ecc_flipped_bit.restype = C.c_uint32
            cls._libs[fmt] = L
        return cls._libs[fmt]

    def __init__(self, fmt, mode, cols, rows, vals, N):
        self.fmt, self.mode, self.N = fmt, mode, int(N)
        self.L = self.lib(fmt)
        self.ctx = self.L.ref_create(mode.encode())
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        self.nnz = len(vals)
        self.mat = self.L.ref_matrix_create(self.ctx, _p(cols, u32p), _p(rows, u32p), _p(vals, f64p), self.N, self.nnz)

    def stored_words(self):
        if self.fmt == CSR:
            c = np.empty(self.nnz, dtype=np.uint32)
            v = np.empty(self.nnz, dtype=np.float64)
            self.L.ref_matrix_read(self.mat, c.ctypes.data, None, v.ctypes.data)
            w = np.empty((self.nnz, 3), dtype=np.uint32)
            w[:, :2] = v.view(np.uint32).reshape(-1, 2)
            w[:, 2] = c
            return w
        w = np.empty((self.nnz, 4), dtype=np.uint32)
        self.L.ref_matrix_read(self.mat, w.ctypes.data, None, None)
        return w

    def rowptr(self):
        rp = np.empty(self.N + 1, dtype=np.uint32)
        self.L.ref_matrix_read(self.mat, None, rp.ctypes.data, None)
        return rp

    def flip(self, index, bits):
        for b in bits:
            self.L.ref_flip(self.mat, index, int(b))

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.full(self.N, np.nan)
        self.L.ref_spmv(self.ctx, self.mat, _p(x, f64p), _p(y, f64p), self.N)
        self.L.ref_flush()
        return y

    def dot(self, a, b):
        return self.L.ref_dot(self.ctx, _p(a, f64p), _p(b, f64p), len(a))

    def calc_xr(self, x, r, p, w, alpha):
        return self.L.ref_calc_xr(self.ctx, _p(x, f64p), _p(r, f64p), _p(p, f64p), _p(w, f64p), alpha, len(x))

    def calc_p(self, p, r, beta):
        self.L.ref_calc_p(self.ctx, _p(p, f64p), _p(r, f64p), beta, len(p))

    def cg(self, b, max_itrs=1000, conv=1e-3):
        n = self.N
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros(n)
        r = np.empty(n)
        p = np.empty(n)
        w = np.empty(n)
        hist = np.zeros(max(1, max_itrs))
        it = self.L.ref_cg(self.ctx, self.mat, _p(b, f64p), _p(x, f64p), _p(r, f64p), _p(p, f64p), _p(w, f64p),
                           n, max_itrs, conv, _p(hist, f64p))
        self.L.ref_flush()
        return it, hist[:it].copy(), x


# ---- helpers that run INSIDE a captured child (tests/_capture.run_captured) ----

def ref_flip_spmv(fmt, mode, mat, index, bits, x, passes=2):
    """flip `bits` of element `index`, then `passes` SpMVs -> ([y...], stored words)"""
    cols, rows, vals, n = mat
    r = Ref(fmt, mode, cols, rows, vals, n)
    r.flip(index, bits)
    ys = [r.spmv(x) for _ in range(passes)]
    return ys, r.stored_words()


def ref_cg(fmt, mode, mat, b, max_itrs=1000, conv=1e-3):
    cols, rows, vals, n = mat
    return Ref(fmt, mode, cols, rows, vals, n).cg(b, max_itrs, conv)


def ref_inject_rand(fmt, mat, seed, kind, flips):
    """srand(seed); inject_bitflip(kind, flips) -> stored words (prints the flip lines)"""
    cols, rows, vals, n = mat
    C.CDLL(None).srand(seed)
    r = Ref(fmt, "none", cols, rows, vals, n)
    r.L.ref_inject_rand(r.ctx, r.mat, kind, flips)
    r.L.ref_flush()
    return r.stored_words()


def ref_encode(fmt, mode, mat):
    cols, rows, vals, n = mat
    r = Ref(fmt, mode, cols, rows, vals, n)
    return r.stored_words(), (r.rowptr() if fmt == CSR else None)


# --------------------------------------------------------- small test matrices --

def laplace5(nx, ny):
    """5-point Laplacian (diag 4, off-diag -1), natural ordering, as full
    symmetric COO triplets sorted by (row, col) -- what cg.cpp:342-418 hands to
    create_matrix after mirroring and sorting."""
    n = nx * ny
    idx = np.arange(n, dtype=np.int64)
    ix, iy = idx % nx, idx // nx
    rows, cols, vals = [], [], []
    for dx, dy, v in ((0, -1, -1.0), (-1, 0, -1.0), (0, 0, 4.0), (1, 0, -1.0), (0, 1, -1.0)):
        ok = (ix + dx >= 0) & (ix + dx < nx) & (iy + dy >= 0) & (iy + dy < ny)
        rows.append(idx[ok])
        cols.append(idx[ok] + dx + dy * nx)
        vals.append(np.full(ok.sum(), v))
    rows, cols, vals = map(np.concatenate, (rows, cols, vals))
    order = np.lexsort((cols, rows))
    return cols[order].astype(np.uint32), rows[order].astype(np.uint32), vals[order], n


def random_spd(n, k, seed):
    """Random symmetric strictly diagonally dominant matrix, ~k off-diagonals per
    row, irrational-looking values (exercises all mantissa bits)."""
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n), max(1, k // 2))
    c = rng.integers(0, n, size=len(r))
    keep = r != c
    r, c = r[keep], c[keep]
    lo, hi = np.minimum(r, c), np.maximum(r, c)
    key = np.unique(lo.astype(np.int64) * n + hi)
    lo, hi = key // n, key % n
    v = -rng.random(len(lo)) - 0.01
    rows = np.concatenate([lo, hi, np.arange(n)])
    cols = np.concatenate([hi, lo, np.arange(n)])
    absrow = np.zeros(n)
    np.add.at(absrow, lo, -v)
    np.add.at(absrow, hi, -v)
    vals = np.concatenate([v, v, absrow + 1.0 + rng.random(n)])
    order = np.lexsort((cols, rows))
    return cols[order].astype(np.uint32), rows[order].astype(np.uint32), vals[order], n


def rhs(n, seed=1):
    return np.random.default_rng(seed).random(n)
