#!/usr/bin/env python3
"""Randomised parity campaign on the GPU: HIP path (through the C ABI) against the CPU
oracle on random matrices, modes, layouts and bit flips -- stored words, SpMV results
(bit for bit, two passes: corrections persist), event streams, and SpMV-in-two-parts.

    python tools/fuzz_parity.py [seconds] [first_seed]

A checker like the tests (it loads oracle/ through tests/_oracle.py); prints one line per
failure with the seed that reproduces it, and a summary."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _oracle import COO, CSR, MODES, OracleMatrix  # noqa: E402

import abft_sparse_cg_amd as amd  # noqa: E402
from abft_sparse_cg_amd import capi  # noqa: E402

NBITS = {CSR: 96, COO: 128}
FNAME = {CSR: "csr", COO: "coo"}


def matrix(rng):
    sizes = [1, 2, 7, 64, 300, 1000, 3000, 6000]
    if os.environ.get("ABFT_FUZZ_BIG") == "1":  # fewer, larger cases (several row blocks per XCD, long sweeps)
        sizes = [20000, 60000, 150000]
    n = int(rng.choice(sizes))
    kind = rng.integers(0, 3)
    rows, cols = [], []
    if n >= 20000:  # vectorised draw for the large cases: k entries per row, duplicates removed
        k = rng.choice([0, 1, 2, 3, 5, 8, 30], size=n)
        r = np.repeat(np.arange(n), k)
        c = rng.integers(0, n, size=len(r))
        key = np.unique(r.astype(np.int64) * n + c)
        rows, cols = [key // n], [key % n]
    else:
        for r in range(n):
            if kind == 0:
                k = int(rng.choice([0, 1, 2, 3, 5, 8]))
            elif kind == 1:
                k = int(rng.choice([0, 0, 1, 4, 30, 200])) if rng.random() < 0.98 else int(rng.integers(1000, 4000))
            else:
                k = int(rng.integers(0, 12))
            k = min(k, n)
            c = np.sort(rng.choice(n, size=k, replace=False))
            rows.append(np.full(k, r))
            cols.append(c)
    rows = np.concatenate(rows) if rows else np.zeros(0)
    cols = np.concatenate(cols) if cols else np.zeros(0)
    vals = rng.standard_normal(len(rows)) * 10.0 ** rng.integers(-3, 4, size=len(rows))
    return cols.astype(np.uint32), rows.astype(np.uint32), vals, n


def bits_equal(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


def one_case(seed):
    rng = np.random.default_rng(seed)
    cols, rows, vals, n = matrix(rng)
    nnz = len(vals)
    fmt = CSR if rng.random() < 0.6 else COO
    mode = str(rng.choice(MODES))
    if os.environ.get("ABFT_FUZZ_ONLY"):  # e.g. coo:constraints -- a campaign on one format and mode
        f, mode = os.environ["ABFT_FUZZ_ONLY"].split(":")
        fmt = COO if f == "coo" else CSR
    layout = str(rng.choice(["stream", "panels", "sweep", "slice", "slice", "auto"]))
    os.environ["ABFT_HIP_SLICE_ROWS"] = str(int(rng.choice([16, 64, 256, 1024])))
    os.environ["ABFT_HIP_SLICE_LAG"] = str(int(rng.choice([0, 1, 2, 3])))
    os.environ["ABFT_HIP_SWEEP_RPT"] = str(int(rng.choice([8, 16])))
    os.environ["ABFT_HIP_SWEEP_LAG"] = str(int(rng.choice([0, 1, 2, 3])))
    os.environ["ABFT_HIP_LAYOUT"] = layout
    os.environ["ABFT_HIP_PANEL_WIDTH"] = str(int(rng.choice([16, 100, 257, 4096])))
    os.environ["ABFT_HIP_PANEL_CHUNK"] = str(int(rng.choice([0, 1, 2, 3])))
    # round 4, COO panel layout: all panels in one launch paced by the per-XCD board (lag > 0; overrides the chunking),
    # workgroups that take several groups in turn, and the opt-in kernels (producer / consumer waves, cold paths out of
    # the hot loop, the x prefetch)
    os.environ["ABFT_HIP_PANEL_LAG"] = str(int(rng.choice([0, 0, 1, 2, 3])))
    os.environ["ABFT_HIP_PANEL_GRID"] = str(int(rng.choice([1, 2, 3, 1000000])))
    kern = int(rng.integers(0, 4))
    os.environ["ABFT_HIP_COO_PC"] = "1" if kern == 1 else "0"
    os.environ["ABFT_HIP_COO_LEAN"] = "1" if kern == 2 else "0"
    os.environ["ABFT_HIP_PANEL_XPF"] = str(int(rng.integers(0, 2)))
    flips = []
    if nnz and (mode not in ("none", "constraints") or rng.random() < 0.7):
        near = int(rng.integers(0, nnz))
        for _ in range(int(rng.integers(0, 4))):
            idx = int(rng.integers(0, nnz))
            if rng.random() < 0.3:  # neighbours in the caller's order (the constraints checks compare those)
                idx = min(nnz - 1, near + int(rng.integers(0, 3)))
            nb = 1 if rng.random() < 0.7 else 2
            if fmt == COO and rng.random() < 0.5:
                # low column bits: the element lands in another output of the vector (the
                # reference scatters there; undetected in none / as a double flip in sec7, sec8)
                hi_bit = max(2, int(n).bit_length())
                nb = min(nb if mode in ("none", "constraints") else 2, hi_bit)
                flips.append((idx, [int(b) for b in rng.choice(hi_bit, size=nb, replace=False)]))
            else:
                flips.append((idx, [int(b) for b in rng.choice(NBITS[fmt], size=nb, replace=False)]))
    x = rng.standard_normal(n)
    seen = []
    o = OracleMatrix(fmt, mode, cols, rows, vals, n)
    ctx = amd.HIPContext(mode, FNAME[fmt], on_event=lambda ev, fatal: seen.append((list(ev), fatal)))
    what = "seed %d: n=%d nnz=%d %s %s layout=%s flips=%s" % (seed, n, nnz, FNAME[fmt], mode, layout, flips)
    try:
        A = ctx.create_matrix(cols, rows, vals, n, nnz)
        if not np.array_equal(ctx.stored_words(A), o.stored_words()):
            return what + " : stored words differ"
        split = fmt == CSR and rng.random() < 0.5 and n > 2
        if split:
            lo = int(rng.integers(0, n))
            ctx.set_interior(A, lo, int(rng.integers(lo, n + 1)))
        for i, b in flips:
            o.inject(i, b)
            ctx.inject_at(A, i, b)
        vx, vy = ctx.create_vector(n), ctx.create_vector(n)
        ctx.upload(vx, x)
        for p in range(2):
            ctx.upload(vy, np.full(n, np.nan))
            seen.clear()
            if split:
                ctx.spmv(A, vx, vy, capi.PART_INTERIOR)
                ctx.spmv(A, vx, vy, capi.PART_BOUNDARY)
            else:
                ctx.spmv(A, vx, vy)
            y = ctx.download(vy)
            ev = [e for evs, _ in seen for e in evs]
            fatal = any(f for _, f in seen)
            want = o.spmv(x)
            oev, ofatal = o.events()
            if (sorted(ev), fatal) != (sorted(oev), ofatal) and not (fatal and ofatal and ev[:1] == oev[:1]):
                return what + " : pass %d events %s fatal=%s, oracle %s fatal=%s" % (p, ev, fatal, oev, ofatal)
            if fatal:
                return None  # the reference stops here
            if not bits_equal(y, want):
                bad = np.nonzero(np.asarray(y).view(np.uint64) != np.asarray(want).view(np.uint64))[0]
                return what + " : pass %d y differs at rows %s" % (p, bad[:5])
        return None
    finally:
        ctx.close()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, done, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        try:
            msg = one_case(seed)
        except Exception as e:  # noqa: BLE001
            msg = "seed %d: exception %r" % (seed, e)
        if msg:
            bad += 1
            print("FAIL " + msg, flush=True)
        done += 1
        seed += 1
        if done % 50 == 0:
            print("... %d cases, %d failures, %.0f s" % (done, bad, time.time() - t0), flush=True)
    print("fuzz: %d cases, %d failures" % (done, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
