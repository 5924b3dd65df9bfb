#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REFERENCE build.

Run in the container that has /root/reference, after `make -C oracle`:

    python tests/golden/make_golden.py

It drives oracle/_ref/libref_{csr,coo}.so (the reference's own CPUContext
classes compiled with -fno-strict-aliasing, see oracle/Makefile) and
oracle/_ref/cg-{csr,coo}-ref (the reference driver) and writes

    tests/golden/ecc.json       masks, known-answer encodes, decode table
    tests/golden/kernels.npz    small matrices, x, expected y per mode, CG histories
    tests/golden/flips.json     injected-flip cases -> exit code, stdout, y (hex)
    tests/golden/lap64.mtx      64x64 5-pt Laplacian, lower triangle (the CLI input)
    tests/golden/cli.json       reference CLI transcripts on lap64.mtx

Only data is written: inputs and the reference's outputs.  Nothing here is used
at product run time; the GPU box never sees /root/reference.
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from _capture import run_captured  # noqa: E402
from _oracle import (COO, CSR, FMT_NAME, MODES, Ref, have_ref, laplace5, random_spd, ref_cg, ref_encode,  # noqa: E402
                     ref_exe, ref_flip_spmv, rhs)

u32p = C.POINTER(C.c_uint32)
NW = {CSR: 3, COO: 4}
EW = {CSR: 2, COO: 0}


def hexd(a):
    return [format(int(v), "016x") for v in np.asarray(a, dtype=np.float64).view(np.uint64)]


def ref_syn(fmt, w):
    w = np.ascontiguousarray(w, dtype=np.uint32)
    return Ref.lib(fmt).ref_ecc_syndrome(w.ctypes.data_as(u32p))


def ref_par(fmt, w):
    w = np.ascontiguousarray(w, dtype=np.uint32)
    return Ref.lib(fmt).ref_ecc_parity(w.ctypes.data_as(u32p))


def ref_encode_via_matrix(fmt, mode, col, row, value_bits):
    """Encode one element by building a 1-element matrix in the reference."""
    val = np.array([value_bits], dtype=np.uint64).view(np.float64)
    n = max(col, row) + 1
    r = Ref(fmt, mode, np.array([col], np.uint32), np.array([row], np.uint32), val, n)
    return int(r.stored_words()[0][EW[fmt]])


def gen_ecc():
    out = {}
    rng = np.random.default_rng(2024)
    for fmt in (CSR, COO):
        name = FMT_NAME[fmt]
        masks = np.zeros((7, NW[fmt]), dtype=np.uint64)
        single = []
        for w in range(NW[fmt]):
            for b in range(32):
                e = np.zeros(NW[fmt], dtype=np.uint32)
                e[w] = 1 << b
                s = ref_syn(fmt, e)
                for p in range(1, 8):
                    if (s >> (32 - p)) & 1:
                        masks[p - 1, w] |= 1 << b
                single.append({"bit": 32 * w + b, "syndrome": s, "parity": ref_par(fmt, e),
                               "decoded": Ref.lib(fmt).ref_ecc_flipped_bit(s) if s else None})
        decode = {str(h): Ref.lib(fmt).ref_ecc_flipped_bit(sum(((h >> (p - 1)) & 1) << (32 - p) for p in range(1, 8)))
                  for h in range(1, 128)}
        kats = []
        fixed = [(0x4010000000000000, 0x000000, 0), (0xBFF0000000000000, 0x000001, 1),
                 (0xBFF3C0CA428C59FB, 0xABCDEF, 0x123456), (0x0, 0xFFFFFF, 0xFFFFFF),
                 (0x01A56E1FC2F8F359, 0x003039, 7), (0x400921FB54442D18, 0x98967F, 0x98967F)]
        rnd = [(int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2)), int(rng.integers(0, 2**24)),
                int(rng.integers(0, 2**24))) for _ in range(40)]
        for vb, col, row in fixed + rnd:
            enc = {m: ref_encode_via_matrix(fmt, m, col, row, vb) for m in MODES}
            kats.append({"value_bits": format(vb, "016x"), "col": col, "row": row, "encoded": enc})
        out[name] = {"masks": [[format(int(v), "08x") for v in r] for r in masks], "single_flips": single,
                     "decode": decode, "kats": kats}
    return out


def mats():
    return {"lap9x7": laplace5(9, 7), "rnd80": random_spd(80, 8, seed=5), "lap16": laplace5(16, 16)}


def gen_kernels():
    arrs = {}
    for name, mat in mats().items():
        cols, rows, vals, n = mat
        arrs[name + "_cols"], arrs[name + "_rows"], arrs[name + "_vals"] = cols, rows, vals
        arrs[name + "_n"] = np.array([n])
        x = rhs(n, 11) - 0.5
        b = rhs(n, 1)
        arrs[name + "_x"], arrs[name + "_b"] = x, b
        for fmt in (CSR, COO):
            for mode in MODES:
                key = "%s_%s_%s" % (name, FMT_NAME[fmt], mode)
                code, text, (words, _) = run_captured(ref_encode, fmt, mode, mat)
                assert code == 0
                arrs[key + "_words"] = words
                code, text, ((y,), _) = run_captured(ref_flip_spmv, fmt, mode, mat, 0, [], x, 1)
                assert code == 0 and text == "", (key, code, text)
                arrs[key + "_y"] = y
                code, text, (it, hist, xs) = run_captured(ref_cg, fmt, mode, mat, b)
                assert code == 0
                arrs[key + "_rr"] = hist
                arrs[key + "_xsol"] = xs
    return arrs


def gen_flips():
    cases = []
    name = "rnd80"
    mat = mats()[name]
    cols, rows, vals, n = mat
    x = rhs(n, 11) - 0.5
    rng = np.random.default_rng(5)
    for fmt in (CSR, COO):
        nb = 96 if fmt == CSR else 128
        for mode in ("sed", "sec7", "sec8", "secded", "constraints"):
            picks = []
            for bit in list(range(0, nb, 5)) + [nb - 1, 24 if fmt == COO else 88]:
                picks.append((int(rng.integers(0, len(vals))), [bit]))
            if mode in ("sec8", "secded"):
                for _ in range(6):
                    b = rng.choice(nb, size=2, replace=False)
                    picks.append((int(rng.integers(0, len(vals))), [int(b[0]), int(b[1])]))
            for index, bits in picks:
                code, text, res = run_captured(ref_flip_spmv, fmt, mode, mat, index, bits, x, 2)
                if code not in (0, 1):
                    continue  # reference itself faulted (UB on a wild gather index)
                c = {"matrix": name, "fmt": FMT_NAME[fmt], "mode": mode, "index": index, "bits": bits,
                     "exit": code, "stdout": text}
                if code == 0:
                    w = [int(v) for v in res[1][index]]
                    ecc = mode != "constraints"
                    gather = (w[2] & 0xFFFFFF if ecc else w[2]) if fmt == CSR else w[1]
                    scatter = 0 if fmt == CSR else (w[0] & 0xFFFFFF if ecc else w[0])
                    if gather >= n or scatter >= n:
                        continue  # the reference indexed outside its vectors (UB): not a vector
                    c["y1"], c["y2"] = hexd(res[0][0]), hexd(res[0][1])
                    c["words_after"] = [[int(v) for v in row] for row in res[1][index:index + 1]]
                cases.append(c)
    return cases


def write_lap_mtx(path, nx, ny):
    n = nx * ny
    ents = []
    for i in range(n):
        ix, iy = i % nx, i // nx
        ents.append((i, i, 4.0))
        if ix > 0:
            ents.append((i, i - 1, -1.0))
        if iy > 0:
            ents.append((i, i - nx, -1.0))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n")
        f.write("%% 5-point Laplacian %dx%d, lower triangle (generated; tests/golden/make_golden.py)\n" % (nx, ny))
        f.write("%d %d %d\n" % (n, n, len(ents)))
        for r, c, v in ents:
            f.write("%d %d %.17g\n" % (r + 1, c + 1, v))


def gen_cli(mtx):
    runs = []
    for fmt in (CSR, COO):
        exe = ref_exe(fmt)
        for b in (1, 4):
            for mode in MODES:
                cmd = [exe, "-f", mtx, "-b", str(b), "-t", "cpu", "-m", mode]
                p = subprocess.run(cmd, capture_output=True, text=True)
                out = re.sub(r"time taken = .*", "time taken = <T> ms", p.stdout)
                runs.append({"fmt": FMT_NAME[fmt], "args": ["-b", str(b), "-m", mode], "exit": p.returncode,
                             "stdout": out})
        p = subprocess.run([exe, "--list"], capture_output=True, text=True)
        runs.append({"fmt": FMT_NAME[fmt], "args": ["--list"], "exit": p.returncode, "stdout": p.stdout})
    return runs


def main():
    if not have_ref():
        sys.exit("oracle/_ref is not built: run `make -C oracle` where /root/reference exists")
    with open(os.path.join(HERE, "ecc.json"), "w") as f:
        json.dump(gen_ecc(), f, indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **gen_kernels())
    with open(os.path.join(HERE, "flips.json"), "w") as f:
        json.dump(gen_flips(), f, indent=0)
    mtx = os.path.join(HERE, "lap64.mtx")
    write_lap_mtx(mtx, 64, 64)
    with open(os.path.join(HERE, "cli.json"), "w") as f:
        json.dump(gen_cli(mtx), f, indent=0)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
