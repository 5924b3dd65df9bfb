#!/usr/bin/env python3
"""Randomised call sequences through the C ABI against a numpy model: spmv, dot, calc_xr,
calc_p, copy_vector, map/unmap in random order on a handful of vectors.  What is under test
is the state the library keeps across calls -- the dot fused into the SpMV and the x update
deferred from calc_xr into calc_p must be invisible under every interleaving: vectors
bit-identical to the model whenever they are read, scalars within the reductions' tolerance.

    python tools/fuzz_sequence.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _oracle import CSR, OracleMatrix, laplace5, random_spd  # noqa: E402

import abft_sparse_cg_amd as amd  # noqa: E402
from abft_sparse_cg_amd import capi  # noqa: E402


def bits_equal(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


def one_case(seed):
    rng = np.random.default_rng(seed)
    if os.environ.get("ABFT_FUZZ_SEQ_GRID"):  # e.g. 12,19: every case on that Laplacian (hunting a size-specific failure)
        g, h = (int(v) for v in os.environ["ABFT_FUZZ_SEQ_GRID"].split(","))
        cols, rows, vals, n = laplace5(g, h)
    elif os.environ.get("ABFT_FUZZ_SEQ_MIX") and seed % 2 == 0:  # every other case one of the sizes the failures were seen on
        g, h = [(38, 6), (12, 19), (6, 38), (18, 3)][(seed // 2) % 4]
        cols, rows, vals, n = laplace5(g, h)
    elif rng.random() < 0.5:
        g = int(rng.integers(3, 40))
        cols, rows, vals, n = laplace5(g, int(rng.integers(3, 40)))
    else:
        cols, rows, vals, n = random_spd(int(rng.choice([50, 300, 2000])), int(rng.integers(2, 12)), seed=seed)
    mode = str(rng.choice(["none", "secded"]))
    o = OracleMatrix(CSR, mode, cols, rows, vals, n)
    ctx = amd.HIPContext(mode, "csr")
    NV = 6
    try:
        if os.environ.get("ABFT_FUZZ_SEQ_PROFILE") == "1":  # HIP-event brackets around every kernel: events created and destroyed per context
            ctx.profile(0xF)
        A = ctx.create_matrix(cols, rows, vals, n, len(vals))
        dev = [ctx.create_vector(n) for _ in range(NV)]
        sc = ctx.create_vector(6)  # device scalars of "devstep": {rr, -}, {rr_new, events}, {p.w, events}
        model = [rng.standard_normal(n) for _ in range(NV)]
        for d, m in zip(dev, model):
            ctx.upload(d, m)
        trace = []

        def whose(p, w):
            """a scalar differs: the model's pieces computed again, the oracle's stored matrix against its input, the device's vectors"""
            w2 = o.spmv(model[p])
            oc, orp, ov = o.csr_arrays()
            o2 = OracleMatrix(CSR, mode, cols, rows, vals, n)
            w3 = o2.spmv(model[p])
            dw, dp = ctx.download(dev[w]), ctx.download(dev[p])
            bad = np.nonzero(np.asarray(model[w]).view(np.uint64) != np.asarray(w3).view(np.uint64))[0]
            return ("model again: dot %r; spmv again equals the model's w: %s (dot %r); a fresh oracle's: %s (dot %r, %d entries differ, first %s); "
                    "device w equals model w %s / the fresh oracle's %s; device p equals model p %s; the oracle's arrays intact: %s"
                    % (float(np.dot(model[p], model[w])), bits_equal(w2, model[w]), float(np.dot(model[p], w2)), bits_equal(w3, model[w]),
                       float(np.dot(model[p], w3)), len(bad), bad[:4], bits_equal(dw, model[w]), bits_equal(dw, w3), bits_equal(dp, model[p]),
                       bool(np.array_equal(oc & 0xFFFFFF, cols) and np.array_equal(o2.csr_arrays()[2], ov) and np.array_equal(o2.csr_arrays()[1], orp))))

        watch = os.environ.get("ABFT_FUZZ_SEQ_WATCH") == "1"  # the checker's own matrix compared with its first image after every operation
        image = o.csr_arrays() if watch else None
        last_op = "create"
        for step in range(int(rng.integers(10, 60))):
            if watch:
                now = o.csr_arrays()
                for name, a, b in zip(("cols", "rowptr", "values"), image, now):
                    if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                        d = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
                        return ("seed %d step %d: the ORACLE's %s changed behind %s (n %d, nnz %d, mode %s): words %s were %s are %s"
                                % (seed, step, name, last_op, n, len(vals), mode, d[:8], a.view(np.uint32)[d[:8]], b.view(np.uint32)[d[:8]]))
            op = str(rng.choice(["spmv", "dot", "calc_xr", "calc_p", "copy", "download", "upload", "cgstep", "devstep"]))
            ids = [int(i) for i in rng.permutation(NV)]
            trace.append((op, ids[:4]))
            last_op = "%s %s (after %s)" % (op, ids[:4], trace[-3:-1])
            if op == "spmv":
                a, b = ids[:2]
                ctx.spmv(A, dev[a], dev[b])
                model[b] = o.spmv(model[a])
            elif op == "dot":
                a, b = (ids[0], ids[1]) if rng.random() < 0.8 else (ids[0], ids[0])
                got = ctx.dot(dev[a], dev[b])
                want = float(np.dot(model[a], model[b]))
                tol = 1e-12 * float(np.abs(model[a] * model[b]).sum()) + 1e-300
                if not abs(got - want) <= tol:
                    return "seed %d step %d %s: dot %r vs %r" % (seed, step, trace[-6:], got, want)
            elif op in ("calc_xr", "cgstep"):
                x, r, p, w = ids[:4]
                alpha = float(rng.uniform(-0.5, 0.5))
                if op == "cgstep":  # the CG order: spmv, dot, calc_xr, calc_p back to back
                    ctx.spmv(A, dev[p], dev[w])
                    model[w] = o.spmv(model[p])
                    pw = ctx.dot(dev[p], dev[w])
                    want = float(np.dot(model[p], model[w]))
                    if not abs(pw - want) <= 1e-12 * float(np.abs(model[p] * model[w]).sum()) + 1e-300:
                        return "seed %d step %d %s: fused dot %r vs %r; %s" % (seed, step, trace[-6:], pw, want, whose(p, w))
                got = ctx.calc_xr(dev[x], dev[r], dev[p], dev[w], alpha)
                model[x] = model[x] + alpha * model[p]
                model[r] = model[r] - alpha * model[w]
                want = float(np.dot(model[r], model[r]))
                if not abs(got - want) <= 1e-12 * want + 1e-300:
                    return "seed %d step %d %s: calc_xr %r vs %r" % (seed, step, trace[-6:], got, want)
                if op == "cgstep":
                    beta = float(rng.uniform(-0.5, 0.5))
                    ctx.calc_p(dev[p], dev[r], beta)
                    model[p] = model[r] + beta * model[p]
            elif op == "devstep":
                # round 4: the whole iteration with the scalars on the device (abft_hip_cg_iteration_dev: SpMV + ONE launch
                # for the fold, the r half and the x / p half); alpha and beta are IEEE quotients of what it leaves behind
                x, r, p, w = ids[:4]
                cur = float(rng.uniform(0.5, 2.0))
                ctx.upload(sc, np.array([cur, 0.0, 0.0, 0.0, 0.0, 0.0]))
                base = sc.device_ptr
                capi.check(ctx.L.abft_hip_cg_iteration_dev(ctx.h, A.h, dev[p].h, 0, capi.PART_ALL, dev[x].h, dev[r].h, dev[p].h,
                                                           dev[w].h, base, base + 32, base + 16))
                got = ctx.download(sc)
                model[w] = o.spmv(model[p])
                want_pw = float(np.dot(model[p], model[w]))
                if not abs(got[4] - want_pw) <= 1e-12 * float(np.abs(model[p] * model[w]).sum()) + 1e-300:
                    return "seed %d step %d %s: devstep p.w %r vs %r; %s" % (seed, step, trace[-6:], got[4], want_pw, whose(p, w))
                with np.errstate(all="ignore"):
                    alpha = np.float64(cur) / np.float64(got[4])
                    model[x] = model[x] + alpha * model[p]
                    model[r] = model[r] - alpha * model[w]
                    want_rr = float(np.dot(model[r], model[r]))
                    if np.isfinite(want_rr) and not abs(got[2] - want_rr) <= 1e-12 * want_rr + 1e-300:
                        return "seed %d step %d %s: devstep r.r %r vs %r" % (seed, step, trace[-6:], got[2], want_rr)
                    beta = np.float64(got[2]) / np.float64(cur)
                    model[p] = model[r] + beta * model[p]
            elif op == "calc_p":
                p, r = ids[:2]
                beta = float(rng.uniform(-0.9, 0.9))
                ctx.calc_p(dev[p], dev[r], beta)
                model[p] = model[r] + beta * model[p]
            elif op == "copy":
                a, b = ids[:2]
                ctx.copy_vector(dev[a], dev[b])
                model[a] = model[b].copy()
            elif op == "upload":
                a = ids[0]
                model[a] = rng.standard_normal(n)
                ctx.upload(dev[a], model[a])
            else:
                a = ids[0]
                if not bits_equal(ctx.download(dev[a]), model[a]):
                    return "seed %d step %d %s: vector %d differs" % (seed, step, trace[-8:], a)
            # keep magnitudes bounded
            for k in range(NV):
                if not np.isfinite(model[k]).all() or np.abs(model[k]).max() > 1e100:
                    model[k] = rng.standard_normal(n)
                    ctx.upload(dev[k], model[k])
        for k in range(NV):
            if not bits_equal(ctx.download(dev[k]), model[k]):
                return "seed %d end %s: vector %d differs" % (seed, trace[-8:], k)
        return None
    finally:
        ctx.close()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
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
        if done % 5000 == 0:
            print("... %d sequences, %d failures, %.0f s" % (done, bad, time.time() - t0), flush=True)
    print("fuzz_sequence: %d sequences, %d failures" % (done, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
