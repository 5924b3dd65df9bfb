"""cg.py -- the reference driver's command line (cg.cpp:38-309) for the hip target on one
GPU, through the ctypes mirror of the plugin interface (context.HIPContext):

    python -m abft_sparse_cg_amd.cg -t hip -s laplace5:3162,3162 -m secded -i 200 -c 0

Same flags, defaults and stdout as cg-csr / cg-coo; additions: --format csr|coo (the
reference picks the format by executable), -s/--synthetic, --seed, --flip-at
INDEX:BIT[,BIT...], -q/--quiet.  It drives HIPContext call for call like the C++ driver.
Several GPUs: the C++ executables (host/cg-csr, host/cg-coo under host/mgpu-run or
torch.distributed.run --no-python) are the one multi-process implementation.
"""
import ctypes
import math
import os
import sys
import time

import numpy as np

DEFAULT_MTX = "matrices/shallow_water1/shallow_water1.mtx"

USAGE = """
Usage: %s [OPTIONS]

Options:
  -h  --help                  Print this message
  -b  --num-blocks      B     Number of times to block input matrix
  -c  --convergence     C     Convergence threshold
  -f  --matrix-file     M     Path to matrix-market format file
  -i  --iterations      I     Maximum number of iterations
  -l  --list                  List available implementations
  -m  --mode            MODE  ABFT mode
  -t  --target          TARG  Implementation target
  -x  --inject-bitflip        Inject a random bit-flip into A

  The -l|--list argument will provide a list of tuples that describe
  which implementations are available to be passed to the
  -t|--target and -m|--mode arguments.

  The -x|--inject-bitflip argument optionally takes a number to
  control how many bits to flip, and either INDEX or VALUE to
  restrict the region of bits in the matrix element to target.

Additional options of this build:
      --format          F     csr (default) or coo
  -s  --synthetic       SPEC  Generate the matrix in memory instead of -f:
                              laplace5:NX,NY | random:N,K,SEED | powerlaw:N,SEED
      --seed            N     Seed for the -x draws (default: time)
      --flip-at  I:B[,B...]   Flip the given bit(s) of matrix element I (may be repeated)
  -q  --quiet                 Do not print the per-iteration residual

"""


def fail(msg):
    print(msg)
    raise SystemExit(1)


def parse(argv):
    o = dict(num_blocks=25, max_itrs=1000, conv=0.001, matrix_file=DEFAULT_MTX, synthetic=None, target="cpu",
             mode="none", flips=0, kind="ANY", seed=None, quiet=False, flip_at=None, fmt="csr", list=False)

    def num(s, conv):
        try:
            return conv(s)
        except ValueError:
            return -1

    i = 1
    while i < len(argv):
        a = argv[i]

        def arg(msg):
            nonlocal i
            i += 1
            if i >= len(argv):
                fail(msg)
            return argv[i]

        if a in ("--convergence", "-c"):
            o["conv"] = num(arg("Invalid convergence threshold"), float)
            if o["conv"] < 0:
                fail("Invalid convergence threshold")
        elif a in ("--iterations", "-i"):
            o["max_itrs"] = num(arg("Invalid number of iterations"), int)
            if o["max_itrs"] < 0:
                fail("Invalid number of iterations")
        elif a in ("--list", "-l"):
            o["list"] = True
        elif a in ("--num-blocks", "-b"):
            o["num_blocks"] = num(arg("Invalid number of blocks"), int)
            if o["num_blocks"] < 1:
                fail("Invalid number of blocks")
        elif a in ("--matrix-file", "-f"):
            o["matrix_file"] = arg("Matrix filename required")
        elif a in ("--mode", "-m"):
            o["mode"] = arg("ABFT mode required")
        elif a in ("--target", "-t"):
            o["target"] = arg("Implementation target required")
        elif a in ("--inject-bitflip", "-x"):
            o["flips"] = 1
            while i + 1 < len(argv) and not argv[i + 1].startswith("-"):
                i += 1
                if argv[i] in ("INDEX", "VALUE"):
                    o["kind"] = argv[i]
                else:
                    o["flips"] = num(argv[i], int)
                    if o["flips"] < 1:
                        fail("Invalid bit-flip parameter")
        elif a in ("--synthetic", "-s"):
            o["synthetic"] = arg("Synthetic matrix specification required")
        elif a == "--format":
            o["fmt"] = arg("Format required")
            if o["fmt"] not in ("csr", "coo"):
                fail("Invalid format")
        elif a == "--seed":
            o["seed"] = num(arg("Invalid seed"), int)
            if o["seed"] < 0:
                fail("Invalid seed")
        elif a == "--flip-at":
            try:
                idx, bits = arg("Invalid --flip-at (want INDEX:BIT[,BIT...])").split(":")
                o["flip_at"] = (o["flip_at"] or []) + [(int(idx), [int(b) for b in bits.split(",")])]  # repeatable: one element each
            except ValueError:
                fail("Invalid --flip-at (want INDEX:BIT[,BIT...])")
        elif a in ("--quiet", "-q"):
            o["quiet"] = True
        elif a in ("--help", "-h"):
            sys.stdout.write(USAGE % os.path.basename(argv[0]))
            raise SystemExit(0)
        else:
            fail("Unrecognized argument '%s' (try '--help')" % a)
        i += 1
    return o


def header(o, n, block, nnz):
    print()
    print("implementation        = %s-%s" % (o["target"], o["mode"]))
    print("matrix size           = %u x %u" % (n, n))
    print("matrix block size     = %u x %u" % (block, block))
    print("number of non-zeros   = %u (%.4f%%)" % (nnz, nnz / (float(n) * float(n)) * 100))
    print("maximum iterations    = %u" % o["max_itrs"])
    print("convergence threshold = %g" % o["conv"])
    print()


def bit_range(fmt, kind):
    if fmt == "csr":
        return {"ANY": (0, 96), "VALUE": (0, 64), "INDEX": (64, 96)}[kind]
    return {"ANY": (0, 128), "VALUE": (64, 128), "INDEX": (0, 64)}[kind]


def draw_flips(o, nnz):
    """-> [(index, [bits]), ...]: --flip-at's elements in the order given, or one element drawn in the reference's rand() order"""
    if o["flip_at"] is not None:
        return o["flip_at"]
    if not o["flips"]:
        return []
    libc = ctypes.CDLL(None)
    libc.srand(o["seed"] if o["seed"] is not None else int(time.time()))
    index = libc.rand() % nnz
    lo, hi = bit_range(o["fmt"], o["kind"])
    return [(index, [libc.rand() % (hi - lo) + lo for _ in range(o["flips"])])]


def main(argv=None):
    argv = sys.argv if argv is None else argv
    o = parse(argv)
    from . import MODES
    if o["list"]:
        print("\nRegistered contexts:")
        for m in list(MODES) + ["sec"]:
            print("\thip-%s" % m)
        print()
        return 0
    if o["target"] != "hip" or o["mode"] not in list(MODES) + ["sec"]:
        sys.stderr.write("\nNo implementation found for %s-%s\n\n" % (o["target"], o["mode"]))
        return 1
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        fail("several ranks: run host/cg-csr or host/cg-coo under this launcher (--no-python); "
             "this Python driver is single-GPU")
    return run_single(o)


def load_matrix(o, row0=0, row1=None):
    from . import generators
    if o["synthetic"]:
        try:
            n = generators.dim(o["synthetic"])
        except ValueError:
            fail("Invalid synthetic matrix '%s'" % o["synthetic"])
        cols, rows, vals, n = generators.generate(o["synthetic"], row0, n if row1 is None else row1)
        return cols, rows, vals, n, n
    try:
        cols, rows, vals, n, block = generators.load_mtx(o["matrix_file"], o["num_blocks"])
    except FileNotFoundError:
        fail("Failed to open '%s'" % o["matrix_file"])
    except ValueError as e:
        fail(str(e))
    if row1 is not None:
        m = (rows >= row0) & (rows < row1)
        cols, rows, vals = cols[m], rows[m], vals[m]
    return cols, rows, vals, n, block


def run_single(o):
    from . import HIPContext, generators
    from .context import fdiv, note_threshold
    cols, rows, vals, n, block = load_matrix(o)
    nnz = len(vals)
    ctx = HIPContext(o["mode"], o["fmt"])
    A = ctx.create_matrix(cols, rows, vals, n, nnz)
    del cols, rows, vals
    header(o, n, block, nnz)
    b, x, r, p, w = (ctx.create_vector(n) for _ in range(5))
    ctx.upload(b, generators.reference_rhs(n))
    ctx.upload(x, np.zeros(n))
    for index, bits in draw_flips(o, nnz):
        for bit in bits:
            print("*** flipping bit %d at index %d ***" % (bit, index))
        ctx.inject_at(A, index, bits)
    t0 = time.perf_counter()
    ctx.copy_vector(r, b)
    ctx.copy_vector(p, r)
    rr = ctx.dot(r, r)
    itr = 0
    noted = {}
    note_threshold(rr, o["conv"], noted)
    while itr < o["max_itrs"] and rr > o["conv"]:
        ctx.spmv(A, p, w)
        pw = ctx.dot(p, w)
        alpha = fdiv(rr, pw)
        rr_new = ctx.calc_xr(x, r, p, w, alpha)
        ctx.calc_p(p, r, fdiv(rr_new, rr))
        rr = rr_new
        note_threshold(rr, o["conv"], noted)
        if not o["quiet"]:
            print("iteration %5u :  rr = %12.4f" % (itr, rr))
        itr += 1
    ms = (time.perf_counter() - t0) * 1e3
    print("\nran for %u iterations" % itr)
    print("\ntime taken = %7.2f ms\n" % ms)
    ctx.spmv(A, x, r)
    err = np.abs(ctx.download(b) - ctx.download(r))
    print("total error = %f" % math.sqrt(float((err * err).sum())))
    print("max error   = %f" % (float(err.max()) if n else 0.0))
    print()
    ctx.destroy_matrix(A)
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
