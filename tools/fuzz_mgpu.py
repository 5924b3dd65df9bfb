#!/usr/bin/env python3
"""The C++ drivers (cg-csr cut by row blocks, cg-coo by column blocks) split over several
processes (host/mgpu-run --one-gpu: every rank on GPU 0, collectives staged through the host)
against the same executable in one process, on random Matrix-Market inputs: same iteration
count, rr lines, ECC event lines (global indices) and error report.  Exercises the partition logic on awkward shapes: few rows per rank, empty
rows at block boundaries, dense and diagonal-only matrices.

    python tools/fuzz_mgpu.py [cases] [first_seed]"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "abft_sparse_cg_amd", "host")


def write_mtx(path, rng):
    n = int(rng.choice([3, 4, 7, 16, 50, 200, 600]))
    dens = float(rng.choice([0.0, 0.02, 0.2, 0.6]))
    ent = {}
    for i in range(n):
        for j in range(i):
            if rng.random() < dens:
                ent[(i, j)] = -float(rng.uniform(0.1, 1.0))
    diag = np.ones(n)
    for (i, j), v in ent.items():
        diag[i] += abs(v)
        diag[j] += abs(v)
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n%d %d %d\n" % (n, n, n + len(ent)))
        for i in range(n):
            f.write("%d %d %.17g\n" % (i + 1, i + 1, diag[i]))
        for (i, j), v in sorted(ent.items()):
            f.write("%d %d %.17g\n" % (i + 1, j + 1, v))
    return n, n + 2 * len(ent)


def norm(text):
    text = re.sub(r"time taken = .*", "", text)
    rr = [float(m) for m in re.findall(r"iteration +\d+ :  rr = +([0-9.]+)", text)]
    rest = re.sub(r"iteration +\d+ :  rr = +[0-9.]+\n", "", text)
    rest = re.sub(r"(total error|max error) += +([0-9.]+)", lambda m: m.group(1) + " = " + m.group(2)[:-2], rest)
    return rr, rest.strip()


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for k in range(cases):
            rng = np.random.default_rng(seed + k)
            path = os.path.join(d, "m.mtx")
            n, nnz = write_mtx(path, rng)
            world = int(rng.choice([2, 3, 4]))
            mode = str(rng.choice(["none", "sed", "sec7", "sec8", "secded", "constraints"]))
            fmt = str(rng.choice(["csr", "coo"]))
            blocks = int(rng.choice([1, 1, 3]))
            args = ["-f", path, "-t", "hip", "-m", mode, "-b", str(blocks)]
            if mode not in ("none", "constraints") and rng.random() < 0.6:
                # (single flips in the ECC modes: detected or corrected on whichever rank holds the element)
                args += ["--flip-at", "%d:%d" % (int(rng.integers(0, nnz * blocks)), int(rng.integers(0, 96 if fmt == "csr" else 128)))]
            exe = os.path.join(HOST, "cg-" + fmt)
            one = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
            many = subprocess.run([os.path.join(HOST, "mgpu-run"), str(world), "--one-gpu", "--", exe] + args,
                                  capture_output=True, text=True, timeout=300)
            what = "seed %d: %s n=%d nnz=%d world=%d %s" % (seed + k, fmt, n, nnz, world, " ".join(args[4:]))
            if n * int(args[args.index("-b") + 1]) < world:
                ok = many.returncode == 2 and "ranks for a matrix" in many.stderr  # refused, loudly
            else:
                (rr1, rest1), (rrn, restn) = norm(one.stdout), norm(many.stdout)
                ok = (one.returncode == many.returncode and rest1 == restn and len(rr1) == len(rrn) and
                      all(abs(a - b) <= 1.01e-4 + 1e-9 * abs(a) for a, b in zip(rr1, rrn)))
            if not ok:
                bad += 1
                print("FAIL %s\n--- one (%d)\n%s\n--- many (%d)\n%s\n%s" % (what, one.returncode, one.stdout[-600:],
                      many.returncode, many.stdout[-600:], many.stderr[-400:]), flush=True)
    print("fuzz_mgpu: %d cases, %d failures" % (cases, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
