"""Host-side logic that needs no GPU: the Matrix-Market loader dialect
(cg.cpp:342-418), the glibc-rand right-hand side, the Python CLI's argument
handling (cg.cpp:180-309), and the panel/partition helpers."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MTX = os.path.join(ROOT, "tests", "golden", "lap64.mtx")


@pytest.fixture(scope="module")
def gen():
    sys.path.insert(0, ROOT)
    from abft_sparse_cg_amd import generators  # plain C++ library: no HIP involved
    return generators


def test_loader_mirrors_sorts_and_tiles(gen, tmp_path):
    p = tmp_path / "t.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real symmetric\n% comment\n%another\n"
                 "3 3 4\n1 1 2.5\n2 1 -1\n3 3 7\n3 2 0.5\n")
    cols, rows, vals, n, block = gen.load_mtx(str(p), 1)
    assert (n, block) == (3, 3)
    # off-diagonals mirrored, sorted by (row, col); first integer of a line is the COLUMN (cg.cpp:371-381)
    assert list(zip(rows, cols, vals)) == [(0, 0, 2.5), (0, 1, -1.0), (1, 0, -1.0), (1, 2, 0.5), (2, 1, 0.5), (2, 2, 7.0)]
    cols2, rows2, vals2, n2, block2 = gen.load_mtx(str(p), 3)  # -b 3: the block down the diagonal
    assert (n2, block2, len(vals2)) == (9, 3, 18)
    for j in range(3):
        s = slice(6 * j, 6 * j + 6)
        assert np.array_equal(cols2[s], cols + 3 * j) and np.array_equal(rows2[s], rows + 3 * j)
        assert np.array_equal(vals2[s], vals)


def test_loader_errors(gen, tmp_path):
    with pytest.raises(FileNotFoundError):
        gen.load_mtx(str(tmp_path / "missing.mtx"), 1)
    p = tmp_path / "rect.mtx"
    p.write_text("%%MatrixMarket\n3 4 1\n1 1 1\n")
    with pytest.raises(ValueError, match="not square"):
        gen.load_mtx(str(p), 1)
    p = tmp_path / "short.mtx"
    p.write_text("%%MatrixMarket\n3 3 2\n1 1 1\n")
    with pytest.raises(ValueError, match="Failed to read"):
        gen.load_mtx(str(p), 1)


def test_committed_laplacian_file_is_what_the_generator_makes(gen):
    cols, rows, vals, n, block = gen.load_mtx(MTX, 1)
    c2, r2, v2, n2 = gen.generate("laplace5:64,64")
    assert n == n2 == 4096 and np.array_equal(cols, c2) and np.array_equal(rows, r2) and np.array_equal(vals, v2)


def test_reference_rhs_is_glibc_rand(gen):
    libc = ctypes.CDLL(None)
    for seed in (1, 77):
        libc.srand(seed)
        want = np.array([libc.rand() for _ in range(3000)]) / 2147483647.0
        assert np.array_equal(gen.reference_rhs(3000, seed), want)


def run_cli(args):
    return subprocess.run([sys.executable, "-m", "abft_sparse_cg_amd.cg"] + args, capture_output=True, text=True,
                          cwd=ROOT, timeout=300)


def test_python_cli_argument_handling_without_a_gpu():
    out = run_cli(["--list"])
    assert out.returncode == 0
    assert out.stdout == "\nRegistered contexts:\n" + "".join(
        "\thip-%s\n" % m for m in ("none", "constraints", "sed", "sec7", "sec8", "secded", "sec")) + "\n"
    out = run_cli(["--bogus"])
    assert out.returncode == 1 and out.stdout == "Unrecognized argument '--bogus' (try '--help')\n"
    out = run_cli(["-i", "x"])
    assert out.returncode == 1 and out.stdout == "Invalid number of iterations\n"
    out = run_cli(["-b", "0"])
    assert out.returncode == 1 and out.stdout == "Invalid number of blocks\n"
    out = run_cli(["-c"])
    assert out.returncode == 1 and out.stdout == "Invalid convergence threshold\n"
    out = run_cli(["-t", "cpu"])
    assert out.returncode == 1 and "No implementation found for cpu-none" in out.stderr
    out = run_cli(["-m", "none"])  # default target is the reference's: cpu (cg.cpp:191)
    assert out.returncode == 1 and "No implementation found for cpu-none" in out.stderr
    out = run_cli(["-t", "hip", "-m", "nonsense"])
    assert out.returncode == 1 and "No implementation found for hip-nonsense" in out.stderr
    out = run_cli(["-t", "hip", "-f", "/nonexistent.mtx"])
    assert out.returncode == 1 and out.stdout == "Failed to open '/nonexistent.mtx'\n"
    out = run_cli(["--help"])
    assert out.returncode == 0 and "--inject-bitflip" in out.stdout and "--synthetic" in out.stdout


def test_flip_draw_ranges_follow_the_reference():
    sys.path.insert(0, ROOT)
    from abft_sparse_cg_amd.cg import bit_range
    # CSR: [0,64) value, [64,96) column (CSR/CPUContext.cpp:139-144); COO: [0,64) indices, [64,128) value
    assert bit_range("csr", "ANY") == (0, 96) and bit_range("csr", "VALUE") == (0, 64)
    assert bit_range("csr", "INDEX") == (64, 96)
    assert bit_range("coo", "ANY") == (0, 128) and bit_range("coo", "VALUE") == (64, 128)
    assert bit_range("coo", "INDEX") == (0, 64)


@pytest.mark.parametrize("world", [2, 5, 8])
def test_host_collectives_of_the_cpp_multi_gpu_backend(world):
    """host/comm.cpp (TCP star through rank 0: rendezvous, bcast, all-gather(v), rank-ordered
    all-reduce) between `world` processes started by host/mgpu-run -- no GPU involved."""
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "abft_sparse_cg_amd", "host")
    exe = os.path.join(host, "comm_test")
    if not os.path.exists(exe):
        pytest.skip("host/comm_test not built")
    out = subprocess.run([os.path.join(host, "mgpu-run"), str(world), "--", exe], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout == "ok\n"  # rank 0's stdout is the job's; the other ranks' is discarded


def test_self_launched_ranks_are_reaped_and_their_status_kept():
    """ABFT_HIP_GPUS=N without a launcher (host/comm.cpp self_launch): the process forks the other ranks before
    anything touches a GPU, binds the rendezvous socket once and keeps it, reaps the ranks on every exit() and
    returns a rank's non-zero status instead of hiding it; under a profiler's preloaded library (which has
    initialised the GPU runtime before main) it refuses to fork and says how to launch instead."""
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "abft_sparse_cg_amd", "host")
    exe = os.path.join(host, "comm_test")
    if not os.path.exists(exe):
        pytest.skip("host/comm_test not built")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    ok = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(env, ABFT_HIP_GPUS="3"))
    assert ok.returncode == 0 and ok.stdout.count("ok\n") == 3, ok.stdout + ok.stderr
    bad = subprocess.run([exe], capture_output=True, text=True, timeout=120,
                         env=dict(env, ABFT_HIP_GPUS="3", COMM_TEST_FAIL_RANK="2"))
    assert bad.returncode == 3 and "a rank started by ABFT_HIP_GPUS ended with status 3" in bad.stderr, bad.stderr
    prof = subprocess.run([exe], capture_output=True, text=True, timeout=120,
                          env=dict(env, ABFT_HIP_GPUS="3", ROCP_TOOL_LIBRARIES="/opt/rocm/lib/librocprofiler-sdk-tool.so"))
    assert prof.returncode == 2 and "cannot fork its ranks under a profiler" in prof.stderr and prof.stdout == ""


@pytest.mark.parametrize("world", [3, 8])
def test_self_launching_rank0_that_fails_ends_its_ranks_instead_of_waiting_for_them(world):
    """Rank 0 exits non-zero (a rendezvous that timed out, check(), a fatal ECC event) while the ranks it started
    are blocked in a receive from it: its exit handler must first shut the sockets (descriptors outlive the
    handlers), so that the ranks see the end of the stream and leave, and bound its wait -- not hang in waitpid."""
    import time
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "abft_sparse_cg_amd", "host")
    exe = os.path.join(host, "comm_test")
    if not os.path.exists(exe):
        pytest.skip("host/comm_test not built")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    t0 = time.time()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=60,
                       env=dict(env, ABFT_HIP_GPUS=str(world), COMM_TEST_RANK0_LEAVES="1"))
    assert p.returncode == 2 and time.time() - t0 < 10.0, (p.returncode, time.time() - t0, p.stderr)
    # every other rank noticed (status 3, "lost a peer") and was reaped: nobody is left behind
    assert p.stderr.count("lost a peer") == world - 1 and "ok" not in p.stdout, p.stderr


def test_bench_ranks_agree_on_every_ranks_exit_status():
    """bench.py --gpus N: before deciding to repeat a failed job, every rank of the launcher learns every rank's
    exit status (bench.agree_codes: a gloo all-reduce on the launcher's own store) -- world size 2 on the CPU."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(25000 + os.getpid() % 4000), os.path.join(root, "tests", "_agree_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert p.returncode == 0, p.stdout[-500:] + p.stderr[-1500:]
    import re
    got = sorted(re.findall(r"AGREE rank \d \[[^\]]*\] \[[^\]]*\]", p.stdout))  # (the two ranks' lines may run together)
    assert got == ["AGREE rank 0 [0, 70] [-6, 0]", "AGREE rank 1 [0, 70] [-6, 0]"], p.stdout


def test_threshold_ambiguity_rule():
    """the stop test is flagged exactly when rr is within 1e-12 (relative) of a non-zero threshold"""
    from abft_sparse_cg_amd.context import threshold_ambiguous
    assert threshold_ambiguous(1e-3, 1e-3)
    assert threshold_ambiguous(1e-3 * (1 + 5e-13), 1e-3) and threshold_ambiguous(1e-3 * (1 - 5e-13), 1e-3)
    assert not threshold_ambiguous(1e-3 * (1 + 1e-11), 1e-3) and not threshold_ambiguous(4863.28, 1e-3)
    assert not threshold_ambiguous(0.0, 0.0) and not threshold_ambiguous(1e-300, 0.0)  # -c 0: fixed iteration count
