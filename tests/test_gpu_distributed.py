"""The sharded solver with the REAL engine (HipEngine -> libabft_hip.so) on the
one GPU of the test box.

* world size 1 over the nccl (RCCL) backend: device-memory aliasing, the shared
  stream and the collective calls end to end;
* world size 2 and 3 with every rank on the same GPU, gloo backend and host-staged
  collectives: the multi-rank shard geometry -- padded column layout, halo windows
  vs all-gather, global event indices, fatal stop on every rank -- with the real
  kernels (RCCL refuses two ranks on one device, so this is how several HIP shards
  can be exercised on a one-GPU box).
The multi-rank collective logic itself is also covered on CPU by
test_distributed_gloo.  Every rank is a child process under a timeout."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from _oracle import CSR, OracleMatrix, laplace5, random_spd, rhs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "_gpu_dist_worker.py")


def run_job(backend, world, matrix, mode, flip, fixed=0, env=None):
    """One attempt.  A rank that hangs or dies fails the test with every rank's output
    (a retry here would hide an intermittent hang in the RCCL / hipGraph path)."""
    idx, bit = flip if flip else (-1, 0)
    env = dict(os.environ, **(env or {}))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [subprocess.Popen([sys.executable, WORKER, backend, str(r), str(world), matrix, mode, str(idx),
                               str(bit), str(port), str(fixed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=env)
             for r in range(world)]
    outs, timed_out = [], False
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            timed_out = True
            break
    if timed_out:
        for p in procs:
            if p.poll() is None:
                p.kill()  # exactly the processes started above
        outs = [p.communicate() for p in procs]
    tails = "\n".join("--- rank %d rc=%s\n%s\n%s" % (r, procs[r].returncode, o[-1500:], e[-3000:])
                      for r, (o, e) in enumerate(outs))
    if timed_out:
        pytest.fail("rank job timed out after 240 s\n" + tails)
    lines = [l for l in outs[0][0].splitlines() if l.startswith("RESULT ")]
    if not lines:
        pytest.fail("rank 0 printed no RESULT line\n" + tails)
    return json.loads(lines[-1][7:]), outs[0][0], [p.returncode for p in procs]


def oracle(matrix, mode, flip):
    cols, rows, vals, n = laplace5(48, 48) if matrix == "laplace" else random_spd(1500, 10, seed=3)
    o = OracleMatrix(CSR, mode, cols, rows, vals, n)
    if flip:
        o.inject(flip[0], [flip[1]])
    it, hist, x, fatal = o.cg(rhs(n, 1))
    ev, _ = o.events()
    return it, hist, x, ev, len(vals)


def check(out, stdout, it_o, hist_o, x_o, ev_o):
    assert out["exit"] == 0 and out["it"] == it_o
    assert np.allclose(out["hist"], hist_o, rtol=1e-10, atol=0)
    assert np.abs(np.array(out["x"]) - x_o).max() <= 1e-10 * np.abs(x_o).max()
    assert [tuple(e) for e in out["events"]] == ev_o
    assert out["tot"] < 0.05 and out["mx"] < 0.01
    for k, i, bit in ev_o:  # rank 0 printed each event once, with the reference's text
        assert stdout.count("at index %d\n" % i) == 1


@pytest.mark.parametrize("mode,flip", [("none", None), ("secded", (777, 40)), ("sec7", (5, 70))])
def test_world1_over_rccl_matches_oracle(mode, flip):
    it_o, hist_o, x_o, ev_o, _ = oracle("laplace", mode, flip)
    out, stdout, codes = run_job("nccl", 1, "laplace", mode, flip)
    check(out, stdout, it_o, hist_o, x_o, ev_o)


def test_world1_sed_fatal_exits_with_status_1():
    out, stdout, codes = run_job("nccl", 1, "laplace", "sed", (99, 3))
    assert out["exit"] == 1
    assert [tuple(e) for e in out["events"]] == [(1, 99, 0)]
    assert "[ECC] error detected at index 99\n" in stdout


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("matrix", ["laplace", "random"])
def test_several_hip_shards_on_one_gpu(world, matrix):
    """real kernels, real shard geometry; a flip in the LAST rank's shard"""
    nnz = oracle(matrix, "none", None)[4]
    flip = (nnz - 40, 21)
    it_o, hist_o, x_o, ev_o, _ = oracle(matrix, "secded", flip)
    out, stdout, codes = run_job("gloo", world, matrix, "secded", flip)
    assert codes == [0] * world
    check(out, stdout, it_o, hist_o, x_o, ev_o)
    assert ev_o == [(2, nnz - 40, 21)]
    assert out["windows"] == (matrix == "laplace")


def test_fatal_event_on_one_shard_stops_every_rank():
    out, stdout, codes = run_job("gloo", 2, "laplace", "sed", (7, 70))
    assert out["exit"] == 1 and [tuple(e) for e in out["events"]] == [(1, 7, 0)]
    assert "[ECC] error detected at index 7\n" in stdout


@pytest.mark.parametrize("backend,world", [("nccl", 1), ("gloo", 2)])
@pytest.mark.parametrize("matrix", ["laplace", "random"])
def test_fixed_iteration_loop_with_device_scalars(backend, world, matrix):
    """run_fixed: spmv+dot, calc_xr and calc_p take alpha / beta from device memory;
    after 15 iterations rr and x equal the reference loop's (and the one corrected
    flip is reported, once, when the loop ends)."""
    cols, rows, vals, n = laplace5(48, 48) if matrix == "laplace" else random_spd(1500, 10, seed=3)
    flip = (len(vals) - 40, 21)
    o = OracleMatrix(CSR, "secded", cols, rows, vals, n)
    o.inject(flip[0], [flip[1]])
    it_o, hist_o, x_o, _ = o.cg(rhs(n, 1), max_itrs=15, conv=0.0)
    out, stdout, codes = run_job(backend, world, matrix, "secded", flip, fixed=15)
    assert codes == [0] * world and out["exit"] == 0 and out["it"] == it_o == 15
    assert abs(out["hist"][-1] - hist_o[-1]) <= 1e-10 * hist_o[-1]
    assert np.abs(np.array(out["x"]) - x_o).max() <= 1e-10 * np.abs(x_o).max()
    assert [tuple(e) for e in out["events"]] == [(2, flip[0], flip[1])]


@pytest.mark.parametrize("matrix", ["laplace", "random"])
def test_graph_replay_equals_eager_enqueue(matrix):
    """run_fixed replays a captured hipGraph per iteration (kernels + RCCL calls, here
    forced on at world size 1): bit-identical to enqueueing the same calls eagerly."""
    force = {"ABFT_FORCE_COLLECTIVES": "1"}
    a, _, ca = run_job("nccl", 1, matrix, "secded", None, fixed=15, env=dict(force, ABFT_CG_GRAPH="1"))
    b, _, cb = run_job("nccl", 1, matrix, "secded", None, fixed=15, env=dict(force, ABFT_CG_GRAPH="0"))
    assert ca == cb == [0]
    assert a["graph"] is True and b["graph"] is False
    assert a["hist"] == b["hist"] and a["x"] == b["x"]
