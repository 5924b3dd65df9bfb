"""HIPContext -- the Python mirror of the reference's CGContext plugin interface
(reference CGContext.h:13-36) for the `hip` target: same thirteen operations,
same argument meaning, same error behaviour (ECC / constraint events are
printed with the reference's text; fatal ones end the run with status 1).

Every method is a thin call into libabft_hip.so through the C ABI of
include/abft_hip.h; the C++ HIPContext under host/ binds the same ABI for the
cg-csr / cg-coo executables.
"""
import ctypes as C
import sys

import numpy as np

from . import capi
from .capi import FMT_COO, FMT_CSR, MODE_ID, check


class FatalEvent(SystemExit):
    """A fatal ABFT event: the reference prints the line and calls exit(1)
    (e.g. CSR/CPUContext.cpp:233-234).  Subclasses SystemExit(1) so an
    unhandled one ends the process exactly like that."""

    def __init__(self, events):
        super().__init__(1)
        self.events = events


class Matrix:
    def __init__(self, ctx, handle, fmt, mode, n_out, n_in, nnz):
        self.ctx, self.h, self.fmt, self.mode = ctx, handle, fmt, mode
        self.n_out, self.n_in, self.nnz = n_out, n_in, nnz


class Vector:
    def __init__(self, ctx, handle, n):
        self.ctx, self.h, self.N = ctx, handle, n
        self._dptr = None

    @property
    def device_ptr(self):
        # asked once: the address never changes, and every call of the C function
        # first applies a deferred update (include/abft_hip.h)
        if self._dptr is None:
            self._dptr = capi.load().abft_hip_vector_device_ptr(self.h)
        return self._dptr


class HIPContext:
    """One (format, mode) backend instance on one GPU -- what
    CGContext::create("hip", mode) returns in cg-csr (fmt='csr') or cg-coo
    (fmt='coo')."""

    BITFLIP = {"ANY": capi.FLIP_ANY, "VALUE": capi.FLIP_VALUE, "INDEX": capi.FLIP_INDEX}

    def __init__(self, mode="none", fmt="csr", device=0, on_event=None, rng=None):
        if mode not in MODE_ID:
            # reference CGContext.cpp:20-23
            sys.stderr.write("\nNo implementation found for hip-%s\n\n" % mode)
            raise SystemExit(1)
        self.L = capi.load()
        self.mode, self.mode_id = mode, MODE_ID[mode]
        self.fmt = FMT_CSR if fmt in ("csr", FMT_CSR) else FMT_COO
        self.on_event = on_event
        self.rng = rng  # callable () -> int, stands in for libc rand(); default: libc
        h = C.c_void_p()
        check(self.L.abft_hip_init(device, C.byref(h)))
        self.h = h
        self.event_log = []
        self._evbuf, self._evcap = None, 0
        self.prof_mask = 0
        self._live = []  # matrices and vectors not yet destroyed, in creation order

    def close(self):
        """destroy what the caller left behind (views before their parents: reverse creation
        order), then the context"""
        if self.h:
            for obj in reversed(self._live):
                if obj.h:
                    if isinstance(obj, Matrix):
                        self.L.abft_hip_matrix_destroy(obj.h)
                    else:
                        self.L.abft_hip_vector_destroy(obj.h)
                    obj.h = None
            self._live = []
            self.L.abft_hip_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- matrix -----------------------------------------------------------
    def create_matrix(self, columns, rows, values, N, nnz, n_in=None, index_base=0):
        """reference CGContext.h:15-18.  n_in/index_base make a row-block shard."""
        columns = np.ascontiguousarray(columns, dtype=np.uint32)
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        values = np.ascontiguousarray(values, dtype=np.float64)
        assert len(columns) >= nnz and len(rows) >= nnz and len(values) >= nnz
        n_in = N if n_in is None else n_in
        h = C.c_void_p()
        check(self.L.abft_hip_matrix_create_shard(
            self.h, self.fmt, self.mode_id, columns.ctypes.data_as(capi.u32p), rows.ctypes.data_as(capi.u32p),
            values.ctypes.data_as(capi.f64p), N, n_in, nnz, index_base, C.byref(h)))
        m = Matrix(self, h, self.fmt, self.mode, N, n_in, nnz)
        self._live.append(m)
        return m

    def destroy_matrix(self, mat):
        self._drain()
        check(self.L.abft_hip_matrix_destroy(mat.h))
        mat.h = None
        self._live = [o for o in self._live if o is not mat]

    def stored_words(self, mat):
        """(nnz, 3|4) uint32 image of the stored elements in the caller's order."""
        if mat.fmt == FMT_CSR:
            c = np.empty(mat.nnz, dtype=np.uint32)
            v = np.empty(mat.nnz, dtype=np.float64)
            check(self.L.abft_hip_matrix_read_csr(mat.h, c.ctypes.data, None, v.ctypes.data))
            w = np.empty((mat.nnz, 3), dtype=np.uint32)
            w[:, :2] = v.view(np.uint32).reshape(-1, 2)
            w[:, 2] = c
            return w
        w = np.empty((mat.nnz, 4), dtype=np.uint32)
        check(self.L.abft_hip_matrix_read_coo(mat.h, w.ctypes.data))
        return w

    def rowptr(self, mat):
        rp = np.empty(mat.n_out + 1, dtype=np.uint32)
        check(self.L.abft_hip_matrix_read_csr(mat.h, None, rp.ctypes.data, None))
        return rp

    # ---- vectors ----------------------------------------------------------
    def create_vector(self, N):
        h = C.c_void_p()
        check(self.L.abft_hip_vector_create(self.h, N, C.byref(h)))
        v = Vector(self, h, N)
        self._live.append(v)
        return v

    def view_vector(self, parent, offset, N):
        h = C.c_void_p()
        check(self.L.abft_hip_vector_view(parent.h, offset, N, C.byref(h)))
        v = Vector(self, h, N)
        self._live.append(v)
        return v

    def destroy_vector(self, vec):
        check(self.L.abft_hip_vector_destroy(vec.h))
        vec.h = None
        self._live = [o for o in self._live if o is not vec]

    def map_vector(self, v):
        """-> numpy view of the pinned staging buffer (valid until unmap)."""
        p = capi.f64p()
        check(self.L.abft_hip_vector_map(v.h, C.byref(p)))
        self._drain()
        if v.N == 0:
            return np.empty(0)
        return np.ctypeslib.as_array(p, shape=(v.N,))

    def unmap_vector(self, v, h):
        if v.N:
            check(self.L.abft_hip_vector_unmap(v.h, h.ctypes.data_as(capi.f64p)))

    def copy_vector(self, dst, src):
        check(self.L.abft_hip_vector_copy(dst.h, src.h))

    def upload(self, v, array):
        h = self.map_vector(v)
        h[:] = array
        self.unmap_vector(v, h)

    def download(self, v):
        h = self.map_vector(v)
        out = h.copy()
        return out

    # ---- kernels ----------------------------------------------------------
    def dot(self, a, b):
        r = C.c_double()
        check(self.L.abft_hip_dot(self.h, a.h, b.h, C.byref(r)))
        self._drain_if_pending()
        return r.value

    def calc_xr(self, x, r, p, w, alpha):
        out = C.c_double()
        check(self.L.abft_hip_calc_xr(self.h, x.h, r.h, p.h, w.h, alpha, C.byref(out)))
        self._drain_if_pending()
        return out.value

    def calc_p(self, p, r, beta):
        check(self.L.abft_hip_calc_p(self.h, p.h, r.h, beta))

    def spmv(self, mat, vec, result, part=capi.PART_ALL):
        if part == capi.PART_ALL:
            check(self.L.abft_hip_spmv(self.h, mat.h, vec.h, result.h))
        else:
            check(self.L.abft_hip_spmv_part(self.h, mat.h, vec.h, result.h, part))

    def matrix_info(self, mat):
        """-> (layout: 'stream' | 'panels' | 'sweep' | 'slice', kernel launches per spmv) -- measurement only"""
        lay, n = C.c_int(0), C.c_int(0)
        check(self.L.abft_hip_matrix_info(mat.h, C.byref(lay), C.byref(n)))
        return ("stream", "panels", "sweep", "slice")[lay.value], n.value

    def set_interior(self, mat, row_lo, row_hi):
        """rows [row_lo, row_hi) read nothing a peer still has to send (include/abft_hip.h)"""
        check(self.L.abft_hip_matrix_set_interior(mat.h, row_lo, row_hi))

    def inject_bitflip(self, mat, kind, num_flips):
        """reference CSR/CPUContext.cpp:135-159 / COO/CPUContext.cpp:123-140: the
        1 + num_flips rand() draws happen here on the host, in that order."""
        kind = self.BITFLIP.get(kind, kind)
        rand = self.rng or _libc_rand
        index = rand() % mat.nnz
        if mat.fmt == FMT_CSR:
            start, end = (0, 64) if kind == capi.FLIP_VALUE else (64, 96) if kind == capi.FLIP_INDEX else (0, 96)
        else:
            start, end = (64, 128) if kind == capi.FLIP_VALUE else (0, 64) if kind == capi.FLIP_INDEX else (0, 128)
        bits = []
        for _ in range(num_flips):
            bit = rand() % (end - start) + start
            sys.stdout.write("*** flipping bit %d at index %d ***\n" % (bit, index))
            bits.append(bit)
        self.inject_at(mat, index, bits)
        return index, bits

    def inject_at(self, mat, index, bits):
        b = np.ascontiguousarray(bits, dtype=np.int32)
        check(self.L.abft_hip_inject(mat.h, index, b.ctypes.data_as(capi.i32p), len(b)))

    # ---- events -----------------------------------------------------------
    def synchronize(self):
        check(self.L.abft_hip_synchronize(self.h))

    def drain_events(self):
        """-> ([(kind, index, bit)], fatal) -- raw, nothing printed."""
        if self._evbuf is None:  # sized to the device queue: drain never truncates
            self._evcap = self.L.abft_hip_event_capacity()
            self._evbuf = (capi.Event * self._evcap)()
        n, fatal = C.c_int(0), C.c_int(0)
        check(self.L.abft_hip_drain_events(self.h, self._evbuf, self._evcap, C.byref(n), C.byref(fatal)))
        return [self._evbuf[i].tup() for i in range(n.value)], bool(fatal.value)

    def _drain_if_pending(self):
        if self.L.abft_hip_pending_events(self.h):
            self._drain()

    def _drain(self):
        events, fatal = self.drain_events()
        if not events:
            return
        self.event_log.extend(events)
        if self.on_event is not None:
            self.on_event(events, fatal)
            return
        for k, i, b in events:
            sys.stdout.write(capi.format_event(k, i, b, self.fmt))
        if fatal:
            sys.stdout.flush()
            raise FatalEvent(events)

    # ---- measurement ------------------------------------------------------
    def profile(self, mask=0xF, stride=1):
        """bit k of mask brackets kernel k (capi.K_*) with HIP events, every
        `stride`-th launch of it; 0 = off"""
        self.prof_mask = 0xF if mask is True else int(mask)
        check(self.L.abft_hip_profile_stride(self.h, stride))
        check(self.L.abft_hip_profile_enable(self.h, self.prof_mask))
        check(self.L.abft_hip_profile_reset(self.h))

    def profile_read(self, kernel):
        ms, n = C.c_double(), C.c_long()
        check(self.L.abft_hip_profile_read(self.h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def stream_probe(self, nbytes=1 << 30, reps=10):
        c, r = C.c_double(), C.c_double()
        check(self.L.abft_hip_stream_probe(self.h, nbytes, reps, C.byref(c), C.byref(r)))
        return c.value, r.value

    @property
    def stream(self):
        return self.L.abft_hip_get_stream(self.h)


_libc = None


def _libc_rand():
    global _libc
    if _libc is None:
        _libc = C.CDLL(None)
    return _libc.rand()


def fdiv(a, b):
    """a / b as the reference's C++ computes it (cg.cpp:102, 109): IEEE-754, so a zero
    denominator gives inf / nan instead of Python's ZeroDivisionError"""
    if b == 0.0:
        a = float(a)
        return float("nan") if a == 0.0 or a != a else float("inf") if (a > 0) == (str(float(b))[0] != "-") else float("-inf")
    return a / b


def threshold_ambiguous(rr, conv_threshold):
    """True when rr lies within rounding of the stop test's threshold (reference cg.cpp:94): the two
    reductions are tree sums, ~1e-13 relative from the reference's serial sums, so only then can the
    iteration count differ from the reference's by one ("iteration-count parity")."""
    return conv_threshold > 0.0 and abs(rr - conv_threshold) <= 1e-12 * abs(rr)


def note_threshold(rr, conv_threshold, state):
    """one stderr line per run when the stop test is ambiguous (stdout stays the reference's)"""
    import sys
    if not state.get("noted") and threshold_ambiguous(rr, conv_threshold):
        state["noted"] = True
        sys.stderr.write("note: threshold-ambiguous run: rr = %s is within 1e-12 (relative) of the convergence "
                         "threshold %.17g; the reference's serially summed rr may fall on the other side and run "
                         "one iteration more or fewer\n" % (float(rr).hex(), conv_threshold))


def cg_solve(ctx, A, b, x, r, p, w, max_itrs=1000, conv_threshold=1e-3, on_iteration=None):
    """The reference driver's CG loop, call for call (cg.cpp:87-118)."""
    ctx.copy_vector(r, b)
    ctx.copy_vector(p, r)
    rr = ctx.dot(r, r)
    itr = 0
    noted = {}
    note_threshold(rr, conv_threshold, noted)
    while itr < max_itrs and rr > conv_threshold:
        ctx.spmv(A, p, w)
        pw = ctx.dot(p, w)
        alpha = fdiv(rr, pw)
        rr_new = ctx.calc_xr(x, r, p, w, alpha)
        beta = fdiv(rr_new, rr)
        ctx.calc_p(p, r, beta)
        rr = rr_new
        note_threshold(rr, conv_threshold, noted)
        if on_iteration is not None:
            on_iteration(itr, rr)
        itr += 1
    return itr, rr
