"""The cg-csr / cg-coo executables (C++ host side: CGContext registry, HIPContext,
driver, Matrix-Market loader) against the REFERENCE driver's own transcripts on
the same input (tests/golden/cli.json: reference cg-csr / cg-coo, -t cpu, on
tests/golden/lap64.mtx).  Everything the reference prints is compared: the
report block, every iteration's rr (to the 4 printed decimals, allowing the last
digit to differ by one: the GPU dot products are tree sums), the iteration
count, and the error lines."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "abft_sparse_cg_amd", "host")
G = os.path.join(ROOT, "tests", "golden")
MTX = os.path.join(G, "lap64.mtx")


def exe(fmt):
    p = os.path.join(HOST, "cg-" + fmt)
    if not os.path.exists(p):
        subprocess.check_call(["make", "-C", HOST])
    return p


def run(fmt, args, env=None):
    # the driver's default target is the reference's, "cpu" (cg.cpp:191); these executables
    # register hip-* only, so every run names it
    if "-t" not in args and "--list" not in args and "--bogus" not in args:
        args = ["-t", "hip"] + args
    return subprocess.run([exe(fmt)] + args, capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, **(env or {})))


def numbers_close(a, b, tol):
    return abs(float(a) - float(b)) <= tol


def compare_transcripts(ours, ref):
    ours = re.sub(r"time taken = .*", "time taken = <T> ms", ours).replace("= hip-", "= cpu-")
    lo, lr = ours.split("\n"), ref.split("\n")
    assert len(lo) == len(lr)
    for a, b in zip(lo, lr):
        if a == b:
            continue
        ma, mb = re.match(r"(iteration +\d+ :  rr = +)([0-9.]+)$", a), re.match(r"(iteration +\d+ :  rr = +)([0-9.]+)$", b)
        assert ma and mb and ma.group(1).split() == mb.group(1).split(), (a, b)
        assert numbers_close(ma.group(2), mb.group(2), 1.01e-4 + 1e-10 * float(mb.group(2))), (a, b)


@pytest.mark.parametrize("fmt", ["csr", "coo"])
def test_cli_transcripts_match_reference(fmt):
    runs = [r for r in json.load(open(os.path.join(G, "cli.json"))) if r["fmt"] == fmt]
    assert len(runs) == 13
    for r in runs:
        if r["args"] == ["--list"]:
            out = run(fmt, ["--list"])
            assert out.returncode == 0
            assert out.stdout == r["stdout"].replace("cpu-", "hip-").replace("\thip-secded\n", "\thip-secded\n\thip-sec\n")
            continue
        out = run(fmt, ["-f", MTX, "-t", "hip"] + r["args"])
        assert out.returncode == r["exit"] == 0, out.stdout[-500:] + out.stderr
        compare_transcripts(out.stdout, r["stdout"])


@pytest.mark.parametrize("driver", ["cpp", "python"])
def test_threshold_ambiguous_run_is_flagged(driver):
    """SURVEY 7 'iteration-count parity': the stop test `rr > threshold` (reference cg.cpp:94) can only
    come out differently from the reference's when rr lands within rounding of the threshold (the two
    reductions are tree sums, ~1e-13 relative from the serial sums).  A run in which that happens says so
    on stderr (stdout stays the reference's); drive it with a threshold crafted from a recorded history."""
    def go(args, env=None):
        if driver == "cpp":
            return run("csr", ["-f", MTX, "-b", "1"] + args, env)
        return subprocess.run([sys.executable, "-m", "abft_sparse_cg_amd.cg", "-t", "hip", "-f", MTX, "-b", "1"] + args,
                              capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(os.environ, **(env or {})))
    hist = run("csr", ["-f", MTX, "-b", "1", "-c", "0", "-i", "60"], {"ABFT_CG_HEX": "1"})
    assert hist.returncode == 0
    rrs = [float.fromhex(m.group(1)) for m in re.finditer(r"^rr \d+ (\S+)$", hist.stderr, re.M)]
    assert len(rrs) == 60 and "threshold-ambiguous" not in hist.stderr
    # (rr is not monotone, and the first values lie above the initial b.b ~ n / 3: take an iteration whose rr is
    # lower than everything before it)
    k = next(i for i in range(20, 60) if rrs[i] < min(rrs[:i]) and rrs[i] < 1000.0)
    # threshold = the rr after iteration k, to all its digits: the loop stops there (rr > thr is false), flagged
    amb = go(["-c", "%.17g" % rrs[k], "-i", "60"])
    assert amb.returncode == 0 and "ran for %d iterations" % (k + 1) in amb.stdout, amb.stdout[-300:]
    assert amb.stderr.count("note: threshold-ambiguous run") == 1 and "threshold-ambiguous" not in amb.stdout
    # one part in 1e9 below it: at least one iteration more, nothing to flag
    clear = go(["-c", "%.17g" % (rrs[k] * (1 - 1e-9)), "-i", "60"])
    assert clear.returncode == 0 and "ran for %d iterations" % (k + 1) not in clear.stdout
    assert "threshold-ambiguous" not in clear.stderr
    # the default run on this input is nowhere near its threshold
    plain = go([])
    assert plain.returncode == 0 and "threshold-ambiguous" not in plain.stderr


@pytest.mark.parametrize("fmt", ["csr", "coo"])
def test_run_tests_script_passes(fmt):
    p = subprocess.run([os.path.join(HOST, "run_tests"), exe(fmt)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout
    assert "FAILED" not in p.stdout and p.stdout.count("passed") >= 7 + 1 + 4 * 3 + 1


@pytest.mark.parametrize("fmt,bit,msg", [("csr", 70, "[ECC] corrected bit 70 at index 1234\n"),
                                         ("coo", 70, "[ECC] corrected bit 70 at index 1234\n")])
def test_replayed_flip_is_corrected_once(fmt, bit, msg):
    """-x replay: the correction line appears exactly once (the write-back
    persists) and the solve then equals the fault-free one."""
    clean = run(fmt, ["-f", MTX, "-b", "1", "-m", "secded"])
    hit = run(fmt, ["-f", MTX, "-b", "1", "-m", "secded", "--flip-at", "1234:%d" % bit])
    assert clean.returncode == 0 and hit.returncode == 0
    assert hit.stdout.count(msg) == 1 and hit.stdout.count("[ECC]") == 1
    assert "*** flipping bit %d at index 1234 ***\n" % bit in hit.stdout
    strip = lambda s: re.sub(r"time taken = .*", "", s.replace(msg, "").replace("*** flipping bit %d at index 1234 ***\n" % bit, ""))
    assert strip(hit.stdout) == strip(clean.stdout)


@pytest.mark.parametrize("driver", ["cpp", "python"])
def test_flip_at_may_be_repeated(driver):
    """--flip-at once per element: two elements repaired, each reported once, the solve equal to the clean one"""
    go = (lambda a: run("csr", a)) if driver == "cpp" else (lambda a: run_py(a))  # noqa: E731
    clean = go(["-f", MTX, "-b", "1", "-m", "secded"])
    hit = go(["-f", MTX, "-b", "1", "-m", "secded", "--flip-at", "1234:70", "--flip-at", "77:3"])
    assert clean.returncode == 0 and hit.returncode == 0
    assert hit.stdout.count("[ECC] corrected bit 70 at index 1234\n") == 1 and hit.stdout.count("[ECC] corrected bit 3 at index 77\n") == 1
    assert hit.stdout.index("index 77\n") < hit.stdout.index("[ECC] corrected bit 70")  # events print in index order
    assert hit.stdout.count("*** flipping bit") == 2 and hit.stdout.count("[ECC]") == 2
    tail = lambda t: t[t.index("iteration     1 :"):t.index("time taken")]  # noqa: E731
    assert tail(hit.stdout) == tail(clean.stdout)


def test_sed_detects_and_exits_1():
    out = run("csr", ["-f", MTX, "-b", "1", "-m", "sed", "--flip-at", "77:3"])
    assert out.returncode == 1
    assert out.stdout.endswith("[ECC] error detected at index 77\n")
    out = run("csr", ["-f", MTX, "-b", "1", "-m", "secded", "--flip-at", "77:3,64"])
    assert out.returncode == 1 and out.stdout.endswith("[ECC] double-bit error detected\n")


def test_cli_errors_like_reference():
    assert run("csr", ["-t", "cpu"]).returncode == 1  # not registered in this executable
    out = subprocess.run([exe("csr"), "-f", MTX], capture_output=True, text=True, timeout=60)  # default target = cpu
    assert out.returncode == 1 and "No implementation found for cpu-none" in out.stderr
    out = run("csr", ["-f", "/nonexistent.mtx"])
    assert out.returncode == 1 and out.stdout == "Failed to open '/nonexistent.mtx'\n"
    out = run("csr", ["--bogus"])
    assert out.returncode == 1 and out.stdout == "Unrecognized argument '--bogus' (try '--help')\n"
    out = run("csr", ["-i", "x"])
    assert out.returncode == 1 and out.stdout == "Invalid number of iterations\n"
    assert run("csr", ["--help"]).returncode == 0


def test_synthetic_input_and_fixed_iterations():
    out = run("csr", ["-s", "laplace5:300,300", "-i", "50", "-c", "0", "-q", "-m", "sec8"])
    assert out.returncode == 0
    assert "matrix size           = 90000 x 90000\n" in out.stdout
    assert "number of non-zeros   = 448800 " in out.stdout
    assert "ran for 50 iterations\n" in out.stdout
    assert "iteration " not in out.stdout


def run_py(args):
    env = dict(os.environ, PYTHONPATH=ROOT)
    if "-t" not in args:
        args = ["-t", "hip"] + args
    cmd = [sys.executable, "-m", "abft_sparse_cg_amd.cg"] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)


def test_python_driver_transcripts_match_reference():
    """abft_sparse_cg_amd.cg prints what the reference cg-csr prints on the same input."""
    runs = {tuple(r["args"]): r for r in json.load(open(os.path.join(G, "cli.json"))) if r["fmt"] == "csr"}
    for args in (["-b", "1", "-m", "none"], ["-b", "4", "-m", "secded"]):
        out = run_py(["-f", MTX] + args)
        assert out.returncode == 0, out.stdout[-800:] + out.stderr[-2000:]
        compare_transcripts(out.stdout, runs[tuple(args)]["stdout"])
    out = run_py(["-f", MTX, "-b", "1", "-m", "sed", "--flip-at", "77:3"])
    assert out.returncode == 1, out.stdout[-500:] + out.stderr[-1500:]
    assert out.stdout.rstrip("\n").endswith("[ECC] error detected at index 77")
    assert "*** flipping bit 3 at index 77 ***" in out.stdout


# ---- several processes, one per GPU (host/mgpu-run + comm.cpp): the unchanged driver,
# ---- row-partitioned by the backend.  On a one-GPU box the ranks share GPU 0 and the
# ---- collectives are staged through the host (--one-gpu); RCCL runs at world size 1.

def hex_history(stderr):
    """ABFT_CG_HEX=1: 'rr <iteration> <%a>' lines on stderr -> {iteration: [value per rank that printed]}"""
    hist = {}
    for m in re.finditer(r"^rr (\d+) (\S+)$", stderr, re.M):
        hist.setdefault(int(m.group(1)), []).append(float.fromhex(m.group(2)))
    return hist


def run_ranks(world, args, opts=(), fmt="csr", env=None):
    cmd = [os.path.join(HOST, "mgpu-run"), str(world)] + list(opts) + ["--", exe(fmt)] + args
    # world 1: still the partitioned code path, with the device collectives on RCCL
    env = dict(os.environ, ABFT_HIP_VERBOSE="1", ABFT_CG_HEX="1", **(env or {}))
    env.setdefault("ABFT_CG_OVERLAP_BYTES", "0")  # interior rows beside the exchange at any exchange size
    if world == 1:
        env["ABFT_COMM_FORCE"] = "1"
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)


def split_transcript(text):
    # (RCCL announces itself on stdout when it initialises: version, HIP, ROCm, hostname, library path)
    text = re.sub(r"(RCCL version|HIP version|ROCm version|Hostname|Librccl path) +:.*\n", "", text)
    if text.startswith("\n\nimplementation") or text.startswith("\nimplementation"):
        text = text.lstrip("\n")
    rr = [float(m) for m in re.findall(r"iteration +\d+ :  rr = +([0-9.]+)", text)]
    rest = re.sub(r"iteration +\d+ :  rr = +[0-9.]+\n", "", re.sub(r"time taken = .*", "", text))
    return rr, rest


@pytest.mark.parametrize("world,opts", [(1, ()), (2, ("--one-gpu",)), (3, ("--one-gpu",))])
@pytest.mark.parametrize("mode,flip", [("none", None), ("secded", "1234:70"), ("sec7", "20000:13")])
def test_cpp_driver_row_partitioned(world, opts, mode, flip):
    """same report, same iteration count, rr lines within the reductions' tolerance, the ECC line
    once and with its global index -- against the one-process run of the same executable"""
    args = ["-f", MTX, "-t", "hip", "-m", mode] + (["--flip-at", flip] if flip else [])
    one = run("csr", args, env={"ABFT_CG_HEX": "1"})
    many = run_ranks(world, args, opts)
    assert one.returncode == 0 and many.returncode == 0, many.stdout[-400:] + many.stderr[-800:]
    rr1, rest1 = split_transcript(one.stdout)
    rrn, restn = split_transcript(many.stdout)
    assert len(rr1) == len(rrn) > 50
    assert all(abs(a - b) <= 1.01e-4 + 1e-10 * a for a, b in zip(rr1, rrn))
    # the residual history with all its bits (north_star: within 1e-10 relative), every rank's copy
    h1, hn = hex_history(one.stderr), hex_history(many.stderr)
    assert sorted(h1) == sorted(hn) == list(range(len(rr1)))
    for it in h1:
        assert len(hn[it]) == world and len(set(hn[it])) == 1, it  # all ranks hold the same scalar
        assert abs(hn[it][0] - h1[it][0]) <= 1e-10 * h1[it][0], it
    # everything else the driver prints (header, flip line, [ECC] line, iteration count, errors)
    norm = lambda t: re.sub(r"(total error|max error) += +[0-9.]+", lambda m: m.group(0)[:-2], t).lstrip("\n")  # noqa: E731
    assert norm(rest1) == norm(restn)
    if flip:
        assert many.stdout.count("[ECC] corrected bit %s at index %s\n" % tuple(reversed(flip.split(":")))) == 1
    # the partition each rank reports (ABFT_HIP_VERBOSE): the banded test matrix exchanges halo
    # windows and multiplies its interior rows beside them; RCCL carries the one-rank job
    notes = [l for l in many.stderr.splitlines() if l.startswith("hip backend: rank")]
    assert len(notes) == world
    if world > 1:
        # (the windows themselves: pushed through IPC-mapped device memory, one kernel per exchange)
        assert all("exchange by windows over device memory (IPC)" in l and "interior rows [0,0)" not in l and
                   l.endswith("beside the exchange") for l in notes)
    else:
        assert "over RCCL" in notes[0]


@pytest.mark.parametrize("world", [1, 3])
def test_cpp_driver_row_partitioned_fatal_event_stops_every_rank(world):
    many = run_ranks(world, ["-f", MTX, "-t", "hip", "-m", "sed", "--flip-at", "30000:70"],
                     ("--one-gpu",) if world > 1 else ())
    assert many.returncode == 1
    assert many.stdout.count("[ECC] error detected at index 30000\n") == 1
    assert "ran for" not in many.stdout


def test_cpp_driver_row_partitioned_synthetic_fixed_iterations():
    """the scattered matrix (all-gather of a vector every rank reads everywhere), -c 0"""
    args = ["-t", "hip", "-m", "secded", "-s", "random:16384,12,3", "-c", "0", "-i", "40"]
    one, many = run("csr", args), run_ranks(3, args, ("--one-gpu",))
    assert one.returncode == 0 and many.returncode == 0, many.stderr[-800:]
    rr1, rest1 = split_transcript(one.stdout)
    rrn, restn = split_transcript(many.stdout)
    assert len(rr1) == len(rrn) == 40
    assert all(abs(a - b) <= 1.01e-4 + 1e-9 * a for a, b in zip(rr1, rrn))
    notes = [l for l in many.stderr.splitlines() if l.startswith("hip backend: rank")]
    assert len(notes) == 3 and all("exchange by all-gather" in l and "interior rows [0,0)" in l for l in notes)
    assert "ran for 40 iterations" in many.stdout


def test_run_tests_script_passes_row_partitioned():
    """the reference's black-box acceptance phases (fault-free, sed detects, sec survives, secded
    flags two flips) with every run split over two processes"""
    launcher = "%s 2 --one-gpu -- %s" % (os.path.join(HOST, "mgpu-run"), exe("csr"))
    p = subprocess.run([os.path.join(HOST, "run_tests"), launcher], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout
    assert "FAILED" not in p.stdout and p.stdout.count("passed") >= 7 + 1 + 4 * 3 + 1


def test_cpp_driver_row_partitioned_short_halo_is_not_overlapped(monkeypatch):
    """default threshold: this matrix's halo (a few KB) goes in front of a single SpMV launch"""
    monkeypatch.setenv("ABFT_CG_OVERLAP_BYTES", str(2 << 20))
    args = ["-f", MTX, "-t", "hip", "-m", "secded", "--flip-at", "1234:70"]
    one, many = run("csr", args), run_ranks(3, args, ("--one-gpu",))
    assert one.returncode == 0 and many.returncode == 0, many.stderr[-800:]
    (rr1, rest1), (rrn, restn) = split_transcript(one.stdout), split_transcript(many.stdout)
    assert len(rr1) == len(rrn) and all(abs(a - b) <= 1.01e-4 + 1e-10 * a for a, b in zip(rr1, rrn))
    notes = [l for l in many.stderr.splitlines() if l.startswith("hip backend: rank")]
    assert len(notes) == 3 and not any(l.endswith("beside the exchange") for l in notes)
    assert many.stdout.count("[ECC] corrected bit 70 at index 1234\n") == 1


def test_cpp_driver_row_partitioned_under_torchrun():
    """the launcher the driver of this repository uses for bench.py also starts the C++ executable
    (--no-python): its agent keeps MASTER_PORT for itself, our rendezvous moves one port up"""
    env = dict(os.environ, ABFT_COMM="tcp", ABFT_HIP_DEVICE="0", ABFT_HIP_VERBOSE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--no-python", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29613", exe("csr"), "-t", "hip", "-m", "secded", "-f", MTX]
    many = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    one = run("csr", ["-t", "hip", "-m", "secded", "-f", MTX])
    assert many.returncode == 0 and one.returncode == 0, many.stderr[-1500:]
    (rr1, rest1), (rrn, restn) = split_transcript(one.stdout), split_transcript(many.stdout)
    assert len(rr1) == len(rrn) and all(abs(a - b) <= 1.01e-4 + 1e-10 * a for a, b in zip(rr1, rrn))
    assert sum(l.startswith("hip backend: rank") for l in many.stderr.splitlines()) == 2


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode,flip", [("none", None), ("secded", "1234:70"), ("sec7", "20000:13"), ("none", "777:3")])
def test_cpp_coo_driver_column_partitioned(world, mode, flip):
    """cg-coo across ranks: the matrix cut by blocks of its outputs (columns),
    each rank's elements scattered over the row-major input and reported with their global
    index -- same report, iteration count, rr history (1e-10) and ECC line as one process.
    ("none" + 777:3 flips a low column bit: within a rank's block the product moves to the
    corrupted column as in the reference.)"""
    args = ["-f", MTX, "-t", "hip", "-m", mode] + (["--flip-at", flip] if flip else [])
    one = run("coo", args, env={"ABFT_CG_HEX": "1"})
    many = run_ranks(world, args, ("--one-gpu",), fmt="coo")
    assert one.returncode == 0 and many.returncode == 0, many.stdout[-400:] + many.stderr[-800:]
    (rr1, rest1), (rrn, restn) = split_transcript(one.stdout), split_transcript(many.stdout)
    assert len(rr1) == len(rrn) > 50
    h1, hn = hex_history(one.stderr), hex_history(many.stderr)
    for it in h1:
        assert len(hn[it]) == world and len(set(hn[it])) == 1, it
        assert abs(hn[it][0] - h1[it][0]) <= 1e-10 * h1[it][0], it
    norm = lambda t: re.sub(r"(total error|max error) += +[0-9.]+", lambda m: m.group(0)[:-2], t).lstrip("\n")  # noqa: E731
    assert norm(rest1) == norm(restn)
    notes = [l for l in many.stderr.splitlines() if l.startswith("hip backend: rank")]
    assert len(notes) == world and all(" columns [" in l for l in notes)


@pytest.mark.parametrize("fmt", ["csr", "coo"])
def test_scalar_allreduces_over_the_peer_board_equal_the_collective_layer(fmt):
    """default: the two scalar all-reduces of an iteration go over the board in shared host memory
    (abft_hip_peer_board_*) and the halo windows through outboxes there (abft_hip_peer_exchange_*),
    here with three processes on the one GPU; ABFT_COMM_ALLREDUCE=tcp / ABFT_COMM_EXCHANGE=tcp keep
    them on the host layer.  Both add in rank order: the same bits, iteration by iteration."""
    args = ["-f", MTX, "-t", "hip", "-m", "secded", "--flip-at", "1234:70"]
    board = run_ranks(3, args, ("--one-gpu",), fmt=fmt, env={"ABFT_COMM_ALLREDUCE": "board", "ABFT_COMM_EXCHANGE": "board"})
    layer = run_ranks(3, args, ("--one-gpu",), fmt=fmt, env={"ABFT_COMM_ALLREDUCE": "tcp", "ABFT_COMM_EXCHANGE": "tcp"})
    # round 3, the default: the board in device memory, a copy per rank, the peers' copies mapped over IPC
    # (between processes sharing the one GPU here; between GPUs the pushes are xGMI stores)
    ipc = run_ranks(3, args, ("--one-gpu",), fmt=fmt)
    assert board.returncode == 0 and layer.returncode == 0 and ipc.returncode == 0, board.stderr[-800:] + layer.stderr[-800:] + ipc.stderr[-800:]
    assert "scalar all-reduces over the peer board in device memory (3 ranks, IPC" in ipc.stderr
    assert ipc.stderr.count("exchange by windows over device memory (IPC)") == 3
    assert hex_history(ipc.stderr) == hex_history(board.stderr)
    assert "scalar all-reduces over the peer board (3 ranks" in board.stderr
    assert board.stderr.count("exchange by windows over shared memory") == 3
    assert "peer board" not in layer.stderr and layer.stderr.count("exchange by windows over TCP") == 3
    hb, hl = hex_history(board.stderr), hex_history(layer.stderr)
    assert len(hb) > 50 and hb == hl
    strip = lambda t: re.sub(r"time taken = .*", "", t)  # noqa: E731
    assert strip(board.stdout) == strip(layer.stdout) and "corrected bit" in board.stdout


def test_bench_loop_fused_and_unfused_allreduces_same_bits():
    """--bench at three ranks on the one GPU: everything in the replayed graph (peer board + outboxes);
    the all-reduces in the tails of the reductions (default) or as kernels of their own, same rr bits"""
    args = ["-t", "hip", "-m", "secded", "-s", "laplace5:150,150", "--bench", "4,31", "-q"]
    fused = run_ranks(3, args, ("--one-gpu",))
    apart = run_ranks(3, args, ("--one-gpu",), env={"ABFT_COMM_FUSE_ALLREDUCE": "0"})
    eager = run_ranks(3, args, ("--one-gpu",), env={"ABFT_CG_GRAPH": "0"})
    inline = run_ranks(3, args, ("--one-gpu",), env={"ABFT_CG_EXCHANGE_BESIDE": "1"})  # the exchange on a graph branch of its own
    one = run("csr", args)
    # round 4: by default everything behind the SpMV is ONE launch (abft_hip_cg_iteration_dev: fold + r half +
    # x / p half + both board all-reduces); ABFT_CG_TAIL=0 keeps the three kernels -- same bits, at 3 ranks and at 1
    three_kernels = run_ranks(3, args, ("--one-gpu",), env={"ABFT_CG_TAIL": "0"})
    one_three = run("csr", args, env={"ABFT_CG_TAIL": "0"})
    for p in (fused, apart, eager, inline, one, three_kernels, one_three):
        assert p.returncode == 0, p.stderr[-800:]
    a, b, c, e, d = (bench_line(p.stdout) for p in (fused, apart, eager, inline, one))
    assert a[0] == b[0] == c[0] == e[0] == 3 and a[4] == b[4] == c[4] == e[4]
    assert abs(a[4] - d[4]) <= 1e-10 * d[4]
    assert bench_line(three_kernels.stdout)[4] == a[4] and bench_line(one_three.stdout)[4] == d[4]


def test_run_tests_script_passes_column_partitioned_coo():
    launcher = "%s 2 --one-gpu -- %s" % (os.path.join(HOST, "mgpu-run"), exe("coo"))
    p = subprocess.run([os.path.join(HOST, "run_tests"), launcher], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout
    assert "FAILED" not in p.stdout and p.stdout.count("passed") >= 7 + 1 + 4 * 3 + 1


def bench_line(out):
    m = re.search(r"^bench: ranks (\d+) warmup (\d+) steps (\d+) seconds ([0-9.]+) iterations_per_second ([0-9.]+) rr (\S+)$",
                  out, re.M)
    assert m, out[-600:]
    return int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4)), float.fromhex(m.group(6))


@pytest.mark.parametrize("spec,mode", [("laplace5:300,300", "secded"), ("random:16384,12,3", "sec8")])
def test_bench_mode_device_scalars_and_graph_replay(spec, mode):
    """--bench W,K (CGContextExt::run_fixed): alpha and beta stay on the device, the iteration is
    captured into a hipGraph and replayed.  After W + K iterations rr equals the driver's own
    -c 0 loop to 1e-10; replay and eager enqueue give the same bits; the partitioned path at
    world size 1 (collectives on RCCL inside the captured graph) and at 2 and 3 ranks (one GPU,
    host-staged collectives, each rank generating only its own row block) agree too."""
    loop = run("csr", ["-s", spec, "-m", mode, "-c", "0", "-i", "25"], env={"ABFT_CG_HEX": "1"})
    assert loop.returncode == 0
    want = hex_history(loop.stderr)[24][0]
    base = ["-s", spec, "-m", mode, "--bench", "5,20", "-q"]
    replay = run("csr", base)
    eager = run("csr", base, env={"ABFT_CG_GRAPH": "0"})
    assert replay.returncode == 0 and eager.returncode == 0, replay.stdout[-400:] + replay.stderr[-800:]
    g, w, k, sec, rr = bench_line(replay.stdout)
    assert (g, w, k) == (1, 5, 20) and sec > 0 and abs(rr - want) <= 1e-10 * want
    assert bench_line(eager.stdout)[4] == rr
    # --bench W,K,B: B blocks, each the whole run again (r = b, W untimed + K timed iterations, between two
    # synchronisations), the line carries the median block; rr is the one of a single block, bit for bit
    blocks = run("csr", ["-s", spec, "-m", mode, "--bench", "5,20,3", "-q"])
    assert blocks.returncode == 0 and bench_line(blocks.stdout)[:3] == (1, 5, 20) and bench_line(blocks.stdout)[4] == rr
    m = re.search(r"^bench_blocks: blocks 3 iterations_per_block 25 seconds((?: [0-9.]+){3})$", blocks.stdout, re.M)
    assert m and sorted(float(t) for t in m.group(1).split())[1] == bench_line(blocks.stdout)[3]
    forced = run_ranks(1, ["-t", "hip"] + base)
    assert forced.returncode == 0, forced.stderr[-800:]
    assert abs(bench_line(forced.stdout)[4] - want) <= 1e-10 * want and "over RCCL" in forced.stderr
    # north_star's transport spelled out -- both scalar all-reduces and the exchange as RCCL calls inside the
    # replayed graph (what the start-up test falls back to when the shared-memory board misbehaves): known good
    rccl = run_ranks(1, ["-t", "hip"] + base, env={"ABFT_COMM_ALLREDUCE": "rccl", "ABFT_COMM_EXCHANGE": "rccl"})
    assert rccl.returncode == 0, rccl.stderr[-800:]
    assert abs(bench_line(rccl.stdout)[4] - want) <= 1e-10 * want
    t = re.findall(r"^bench_transport: (.*)$", rccl.stdout, re.M)
    assert len(t) == 1 and " allreduce rccl " in t[0] and "-over-rccl graph 1 ncclCommCount 1" in t[0], t
    t = re.findall(r"^bench_transport: (.*)$", forced.stdout, re.M)
    assert len(t) == 1 and " allreduce device-board-in-kernel-tails " in t[0] and "graph 1 ncclCommCount 1" in t[0], t
    for world in (2, 3):
        many = run_ranks(world, ["-t", "hip"] + base, ("--one-gpu",))
        assert many.returncode == 0, many.stderr[-800:]
        gg, _, _, _, rrn = bench_line(many.stdout)
        assert gg == world and abs(rrn - want) <= 1e-10 * want


def test_config3_cli_sed_x_index_full_size():
    """BASELINE.json configs[2] as it is spelled: cg-csr -t hip -m sed -x INDEX on the 50 M-nnz
    Laplacian through the C++ driver (reference cg.cpp:254-274 -> CSR/CPUContext.cpp:135-159):
    the flipped column word is caught by the first SpMV, the run ends with status 1."""
    out = subprocess.run([exe("csr"), "-t", "hip", "-s", "laplace5:3162,3162", "-m", "sed", "-x", "INDEX", "--seed", "7",
                          "-q"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 1, out.stdout[-600:] + out.stderr[-600:]
    m = re.search(r"\*\*\* flipping bit (\d+) at index (\d+) \*\*\*\n", out.stdout)
    assert m and 64 <= int(m.group(1)) < 96 and 0 <= int(m.group(2)) < 49978572
    assert out.stdout.endswith("[ECC] error detected at index %s\n" % m.group(2))
    assert "number of non-zeros   = 49978572 " in out.stdout and "ran for" not in out.stdout


@pytest.mark.parametrize("fmt", ["csr", "coo"])
def test_run_benchmark_script(fmt):
    """host/run_benchmark (reference run_benchmark:1-28): mean (min / max) of 'time taken' over
    NUM_RUNS runs for every registered implementation."""
    p = subprocess.run([os.path.join(HOST, "run_benchmark"), exe(fmt), "-f", MTX, "-b", "2"], capture_output=True,
                       text=True, timeout=900, env=dict(os.environ, NUM_RUNS="2"))
    assert p.returncode == 0, p.stdout + p.stderr
    assert "across 2 runs" in p.stdout
    rows = re.findall(r"^(hip-\w+) +: +([0-9.]+) ms   \( +([0-9.]+)  /  +([0-9.]+) \)$", p.stdout, re.M)
    assert [r[0] for r in rows] == ["hip-" + m for m in ("none", "constraints", "sed", "sec7", "sec8", "secded", "sec")]
    for _, mean, lo, hi in rows:
        assert 0.0 < float(lo) <= float(mean) <= float(hi)


def test_bench_py_one_gpu_line_carries_the_contract_keys():
    """`python bench.py` at N = 1 on a small matrix: ONE JSON line with the driver's keys, `roofline` (bound, achieved,
    peak, frac, traffic) and `cpu_baseline` (value, unit, cores, kind, sample: mean / min / max of 5 runs, one
    core beside all), the K timed steps as 5 blocks with the median reported, and the like-for-like graph loop."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--spec",
                        "laplace5:400,400", "--cpu-iters", "20", "--no-probe"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-500:] + p.stderr[-1500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["blocks"] == 5 and d["value_min"] <= d["value"] <= d["value_max"] and d["config"]["N"] == 160000
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["runs"] == 5 and c["min"] <= c["value"] <= c["max"]
    assert c["one_core"]["cores"] == 1 and c["one_core"]["runs"] == 5 and "sample" in c
    legs = d["extra_legs"]
    assert legs["config2_graph_loop"]["it_per_s"] > 0 and legs["config2_graph_loop"]["graph_replay"] is True
    assert legs["config2_graph_loop"]["blocks"] == 5 and legs["config2_graph_loop"]["iterations_per_block"] == 3 + 12
    # the Python loop through the C ABI (host scalars) and the C++ graph loop (device scalars) after the same 15
    # iterations from the same start: the same residual to the reductions' tolerance
    assert abs(legs["config2_graph_loop"]["rr_after_last_step"] - d["config"]["rr_after_last_step"]) <= 1e-10 * d["config"]["rr_after_last_step"]
    # the N = 1 base of the matrix north_star's 8-GPU target is quoted on: same loop as the --gpus N lines' extra leg
    c4 = legs["config4_graph_loop"]
    assert c4["N"] == 4194304 and c4["it_per_s"] > 0 and c4["graph_replay"] is True and c4["blocks"] == 5
    assert d["config"]["iterations_per_block"] == 3 + 12 and "traffic_source" in r and "frac_of_measured_copy" not in r
    assert legs["config4_shard1"]["layout"] == "sweep" and legs["config5"]["layout"] == "panels"


def test_bench_py_multi_rank_path_runs_the_cpp_driver():
    """bench.py --gpus N (N > 1 under the launcher) starts host/cg-csr --bench per rank; here the
    same code path with one rank (collectives forced onto RCCL), small matrix, no extra leg."""
    env = dict(os.environ, ABFT_BENCH_SHARDED="1", ABFT_COMM_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(21000 + os.getpid() % 9000))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3",
                        "--spec", "laplace5:300,300", "--mode", "secded", "--no-extras"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-500:] + p.stderr[-1500:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["value"] > 0 and d["config"]["N"] == 90000
    assert d["roofline"]["avg_launch_us"] > 0 and "C++ host over RCCL" in d["config"]["parallelism"]
    # the run says what carried it: one line per rank, with RCCL's own count of its communicator
    assert len(d["transport_by_rank"]) == 1 and d["transport_by_rank"][0].startswith("rank 0 device 0 allreduce ")
    assert "ncclCommCount 1" in d["transport_by_rank"][0] and "first_attempt" not in d


def test_bench_py_under_the_launcher_two_ranks_two_jobs():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per process; here both
    ranks on the one GPU with host-staged collectives): every rank starts cg-csr --bench, twice in a row
    on the same rendezvous port (headline + the extra leg), rank 0 prints one JSON line."""
    env = dict(os.environ, ABFT_COMM="tcp", ABFT_HIP_DEVICE="0", ABFT_BENCH_EXTRA_SPEC="random:16384,12,3")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(23000 + os.getpid() % 6000), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "8", "--warmup", "3", "--spec", "laplace5:200,200", "--mode", "secded"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-800:] + p.stderr[-2500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["N"] == 40000 and d["scaling"] == "strong"
    assert d["extra_legs"]["config4"]["N"] == 16384 and d["extra_legs"]["config4"]["it_per_s"] > 0
    # self-validation of an unattended multi-rank run: transports by rank, and the same W + K iterations by ONE
    # process give the same rr (1e-10) -- for the headline and for the extra leg
    assert [t.split()[1] for t in d["transport_by_rank"]] == ["0", "1"] and "first_attempt" not in d
    for rec in (d, d["extra_legs"]["config4"]):
        assert rec["rr_check"]["ok"] and rec["rr_check"]["rel_diff"] <= 1e-10, rec["rr_check"]


def test_bench_py_failed_first_attempt_is_agreed_on_and_recorded():
    """A rank whose job dies (here: rank 1's first attempt, injected) must not leave the others waiting or turn
    into a quietly different measurement: every rank learns every rank's status (a gloo all-reduce), ALL repeat
    the job once in the conservative form (eager, collectives on the collective layer), and the output line
    carries the failed attempt -- statuses by rank and rank 0's stderr tail -- as `first_attempt`."""
    env = dict(os.environ, ABFT_COMM="tcp", ABFT_HIP_DEVICE="0", ABFT_BENCH_INJECT_FAILURE="1", ABFT_COMM_TIMEOUT="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(24000 + os.getpid() % 5000), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "6", "--warmup", "3", "--spec", "laplace5:150,150", "--mode", "secded", "--no-extras"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-800:] + p.stderr[-2500:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    fa = d["first_attempt"]
    assert fa["returncodes_by_rank"][1] != 0 and fa["returncodes_by_rank"][0] != 0  # rank 0 lost its peer at the rendezvous
    assert "rendezvous timed out" in fa["rank0_stderr_tail"] and "rccl" in fa["second_attempt"]
    assert d["value"] > 0 and d["rr_check"]["ok"] and "enqueued eagerly" in d["config"]["parallelism"]
    assert all(" graph 0 " in t for t in d["transport_by_rank"])


@pytest.mark.parametrize("fmt", ["csr", "coo"])
def test_abft_hip_gpus_starts_the_ranks_itself(fmt):
    """ABFT_HIP_GPUS=N and no launcher: the executable forks the other ranks before touching a GPU
    (here all on GPU 0, host-staged collectives) -- same transcript as one process."""
    args = ["-f", MTX, "-t", "hip", "-m", "secded", "--flip-at", "1234:70"]
    one = run(fmt, args)
    env = dict(os.environ, ABFT_HIP_GPUS="3", ABFT_COMM="tcp", ABFT_HIP_DEVICE="0", ABFT_HIP_VERBOSE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    many = subprocess.run([exe(fmt)] + args, capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0 and many.returncode == 0, many.stdout[-400:] + many.stderr[-1200:]
    (rr1, rest1), (rrn, restn) = split_transcript(one.stdout), split_transcript(many.stdout)
    assert len(rr1) == len(rrn) > 50 and all(abs(a - b) <= 1.01e-4 + 1e-10 * a for a, b in zip(rr1, rrn))
    norm = lambda t: re.sub(r"(total error|max error) += +[0-9.]+", lambda m: m.group(0)[:-2], t).lstrip("\n")  # noqa: E731
    assert norm(rest1) == norm(restn)
    assert len([l for l in many.stderr.splitlines() if l.startswith("hip backend: rank")]) == 3


@pytest.mark.parametrize("fmt", ["csr", "coo"])
def test_four_ranks_and_run_to_run_reproducibility(fmt):
    """G = 4 (SURVEY 8f row 1 asks for G in {1, 2, 4, 8}; four processes share the one GPU here), and
    SURVEY 8e's determinism: fixed reduction shapes per shard + sums in rank order => two runs of the
    same job give the same residual history bit for bit."""
    args = ["-f", MTX, "-t", "hip", "-m", "secded", "--flip-at", "4321:17"]
    one = run(fmt, args, env={"ABFT_CG_HEX": "1"})
    a = run_ranks(4, args, ("--one-gpu",), fmt=fmt)
    b = run_ranks(4, args, ("--one-gpu",), fmt=fmt)
    assert one.returncode == 0 and a.returncode == 0 and b.returncode == 0, a.stderr[-800:]
    h1, ha, hb = hex_history(one.stderr), hex_history(a.stderr), hex_history(b.stderr)
    assert sorted(ha) == sorted(hb) == sorted(h1) and len(h1) > 50
    for it in h1:
        assert len(ha[it]) == 4 and len(set(ha[it])) == 1 and ha[it] == hb[it], it  # identical across ranks and runs
        assert abs(ha[it][0] - h1[it][0]) <= 1e-10 * h1[it][0], it
    assert a.stdout.count("[ECC] corrected bit 17 at index 4321\n") == 1
    (_, rest1), (_, resta) = split_transcript(one.stdout), split_transcript(a.stdout)
    norm = lambda t: re.sub(r"(total error|max error) += +[0-9.]+", lambda m: m.group(0)[:-2], t).lstrip("\n")  # noqa: E731
    assert norm(rest1) == norm(resta)


# ---- BASELINE.json configs[3] end to end, at its size: cg-csr -t hip -m secded, random 2^22 x 24 (104.9 M non-zeros),
# ---- row-partitioned.  The GPU boxes of this pool allow at most 6 processes on a card at once (gpurun's process
# ---- guard: a run with 6 ranks + this test process was killed, gpurun_out/r4/gpu_all_2.log), so the partitioned
# ---- ITERATION runs here with 5 ranks sharing the one GPU beside the test process -- the same code at any rank
# ---- count (what is specific to 8 -- rendezvous, board slots, the planner -- has its own tests: host collectives and
# ---- the failing-rank-0 exit at 8 processes in test_host_logic.py, the planner at 8 in test_partition.py, the board
# ---- and the window kernels with 8 contexts in test_gpu_peer_board.py, shards 0 / 3 / 7 of 8 in test_gpu_fullsize.py).
CONFIG4 = "random:4194304,24,1"
RANKS_ONE_GPU = 5


def test_config4_end_to_end_row_partitioned_at_full_size():
    """(a) --bench 3,8: rr after 11 iterations within 1e-10 of the one-process run of the same loop, one
    bench_transport line per rank (scalar all-reduces over the IPC device-memory board between the processes,
    the 33.5 MB all-gather staged through the host: RCCL refuses two ranks on one device).
    (b) the reference loop (-c 0 -i 6) with one flipped bit in the first rank's row block and one in the last
    rank's: the one-process run's [ECC] lines, with their global element indices, once each; same residuals."""
    base = ["-t", "hip", "-m", "secded", "-s", CONFIG4, "-q"]
    one = run("csr", base + ["--bench", "3,8"])
    many = run_ranks(RANKS_ONE_GPU, base + ["--bench", "3,8"], ("--one-gpu",))
    assert one.returncode == 0 and many.returncode == 0, many.stdout[-600:] + many.stderr[-1500:]
    g1, _, _, _, rr1 = bench_line(one.stdout)
    gn, w, k, sec, rrn = bench_line(many.stdout)
    assert (g1, gn, w, k) == (1, RANKS_ONE_GPU, 3, 8) and sec > 0
    assert abs(rrn - rr1) <= 1e-10 * rr1, (rrn, rr1)
    t = re.findall(r"^bench_transport: (.*)$", many.stdout, re.M)
    assert [l.split()[1] for l in t] == [str(r) for r in range(RANKS_ONE_GPU)], t
    assert all(" allreduce device-board" in l and " exchange allgather-over-tcp " in l for l in t), t
    assert "number of non-zeros   = 104857298 " in many.stdout
    notes = [l for l in many.stderr.splitlines() if l.startswith("hip backend: rank")]
    assert len(notes) == RANKS_ONE_GPU and all("exchange by all-gather" in l for l in notes)

    nnz = 104857298
    flips = ["--flip-at", "1000:70", "--flip-at", "%d:13" % (nnz - 1000)]
    loop = ["-t", "hip", "-m", "secded", "-s", CONFIG4, "-c", "0", "-i", "6"] + flips
    one = run("csr", loop, env={"ABFT_CG_HEX": "1"})
    many = run_ranks(RANKS_ONE_GPU, loop, ("--one-gpu",))
    assert one.returncode == 0 and many.returncode == 0, many.stdout[-600:] + many.stderr[-1500:]
    for out in (one.stdout, many.stdout):
        assert out.count("[ECC] corrected bit 70 at index 1000\n") == 1
        assert out.count("[ECC] corrected bit 13 at index %d\n" % (nnz - 1000)) == 1 and out.count("[ECC]") == 2
    # the flipped elements live on the first and on the last rank
    first = re.search(r"rank 0 of \d+: rows \[0,\d+\), (\d+) non-zeros from element 0,", many.stderr)
    last = re.search(r"rank %d of \d+: rows \[\d+,4194304\), (\d+) non-zeros from element (\d+)," % (RANKS_ONE_GPU - 1), many.stderr)
    assert first and last and int(first.group(1)) > 1000 and int(last.group(2)) < nnz - 1000
    h1, hn = hex_history(one.stderr), hex_history(many.stderr)
    assert sorted(h1) == sorted(hn) == list(range(6))
    for it in h1:
        assert len(hn[it]) == RANKS_ONE_GPU and len(set(hn[it])) == 1, it
        assert abs(hn[it][0] - h1[it][0]) <= 1e-10 * h1[it][0], it
    (_, rest1), (_, restn) = split_transcript(one.stdout), split_transcript(many.stdout)
    norm = lambda t: re.sub(r"(total error|max error) += +[0-9.]+", lambda m: m.group(0)[:-2], t).lstrip("\n")  # noqa: E731
    assert norm(rest1) == norm(restn)


def test_bench_py_under_the_launcher_at_full_size():
    """bench.py --gpus 5 exactly as the driver launches it (torch.distributed.run, one rank per process), all five
    ranks on the one GPU with the bandwidth collectives host-staged: the headline (config 2, full size: halo windows
    pushed through IPC device memory, the iteration replayed as a graph) and the extra leg on BASELINE.json configs[3]
    at ITS size -- both self-validated against one process (rr_check), a transport line per rank each."""
    env = dict(os.environ, ABFT_COMM="tcp", ABFT_HIP_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(RANKS_ONE_GPU),
           "--master-addr", "127.0.0.1", "--master-port", str(25000 + os.getpid() % 4000), os.path.join(ROOT, "bench.py"),
           "--gpus", str(RANKS_ONE_GPU), "--steps", "4", "--warmup", "3"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-800:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == RANKS_ONE_GPU and d["config"]["N"] == 9998244 and d["config"]["nnz"] == 49978572 and d["value"] > 0
    assert d["blocks"] == 5 and d["value_min"] <= d["value"] <= d["value_max"] and d["config"]["iterations_per_block"] == 3 + 4
    x = d["extra_legs"]["config4"]
    assert x["N"] == 4194304 and x["nnz"] == 104857298 and x["it_per_s"] > 0 and x["iterations_per_block"] == 7
    for rec in (d, x):
        assert "first_attempt" not in rec
        assert [t.split()[1] for t in rec["transport_by_rank"]] == [str(r) for r in range(RANKS_ONE_GPU)]
        assert rec["rr_check"]["ok"] and rec["rr_check"]["rel_diff"] <= 1e-10, rec["rr_check"]
    assert all(" graph 1 " in t for t in d["transport_by_rank"])  # config 2: board + windows, captured and replayed
