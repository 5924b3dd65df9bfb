#!/usr/bin/env python3
"""The sharded solver's partition / exchange / reduction / event logic on CPU (gloo, the
tests' CPU stand-in engine) over random matrices, world sizes and flips, against the
single-process oracle.  Reuses tests/test_distributed_gloo.py's harness.

    python tools/fuzz_sharded_cpu.py [cases] [first_seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_distributed_gloo as T  # noqa: E402
from _oracle import laplace5, random_spd  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = 0
    for k in range(cases):
        rng = np.random.default_rng(seed + k)
        if rng.random() < 0.5:
            cols, rows, vals, n = laplace5(int(rng.integers(4, 30)), int(rng.integers(4, 30)))
        else:
            cols, rows, vals, n = random_spd(int(rng.choice([40, 150, 500])), int(rng.integers(2, 10)), seed=seed + k)
        world = int(rng.choice([2, 3, 4]))
        mode = str(rng.choice(["none", "secded", "sec7"]))
        flip = None
        if mode != "none" and rng.random() < 0.7:
            flip = (int(rng.integers(0, len(vals))), [int(rng.integers(0, 96))])
        bounds = T.uneven_bounds(rows, n, world)
        if any(bounds[g + 1] <= bounds[g] for g in range(world)):
            continue
        fixed = int(rng.integers(3, 15)) if rng.random() < 0.4 else 0
        what = "seed %d: n=%d nnz=%d world=%d %s flip=%s fixed=%d" % (seed + k, n, len(vals), world, mode, flip, fixed)
        try:
            o = T.OracleMatrix(T.CSR, mode, cols, rows, vals, n)
            if flip:
                o.inject(*flip)
            if fixed:
                it_s, hist_s, x_s, _ = o.cg(T.rhs(n, 1), max_itrs=fixed, conv=0.0)
            else:
                it_s, hist_s, x_s, _ = o.cg(T.rhs(n, 1))
            ev_s, _ = o.events()
            code, it, hist, x, tot, mx, events, _ = T.run_case(world, (cols, rows, vals, n, bounds, mode, flip, fixed))
            ok = code == 0 and it == it_s and events == ev_s and np.abs(x - x_s).max() <= 1e-9 * max(np.abs(x_s).max(), 1e-300)
            ok = ok and abs(hist[-1] - hist_s[-1]) <= 1e-9 * max(hist_s[-1], 1e-300)
        except Exception as e:  # noqa: BLE001
            ok = False
            what += " exception %r" % (e,)
        if not ok:
            bad += 1
            print("FAIL " + what, flush=True)
    print("fuzz_sharded_cpu: %d cases, %d failures" % (cases, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
