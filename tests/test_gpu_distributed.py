"""The sharded solver with the REAL engine (HipEngine -> libabft_hip.so) on the
one GPU of the test box: world_size 1 over the nccl (RCCL) backend, which
exercises the device-memory aliasing, the shared stream and the collective calls
end to end.  Multi-rank logic is covered on CPU by test_distributed_gloo.

Each case runs in a child process (tests/_gpu_dist_worker.py) under a timeout:
RCCL bootstrap inside a long-lived pytest process was seen to stall once, and a
stall there must cost one test, not the whole run."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from _oracle import CSR, OracleMatrix, laplace5, rhs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_worker(mode, flip):
    idx, bit = flip if flip else (-1, 0)
    last = None
    for attempt in range(2):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        try:
            p = subprocess.run([sys.executable, os.path.join(HERE, "_gpu_dist_worker.py"), mode, str(idx), str(bit),
                                str(port)], capture_output=True, text=True, timeout=240)
        except subprocess.TimeoutExpired as e:
            last = "timeout: %s" % e
            continue
        lines = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if lines:
            return json.loads(lines[-1][7:]), p.stdout
        last = p.stdout[-2000:] + p.stderr[-2000:]
    pytest.fail("worker did not finish: %s" % last)


@pytest.mark.parametrize("mode,flip", [("none", None), ("secded", (777, 40)), ("sec7", (5, 70))])
def test_sharded_engine_world1_matches_oracle(mode, flip):
    cols, rows, vals, n = laplace5(48, 48)
    b = rhs(n, 1)
    o = OracleMatrix(CSR, mode, cols, rows, vals, n)
    if flip:
        o.inject(flip[0], [flip[1]])
    it_o, hist_o, x_o, _ = o.cg(b)
    ev_o, _ = o.events()
    out, stdout = run_worker(mode, flip)
    assert out["exit"] == 0 and out["it"] == it_o
    assert np.allclose(out["hist"], hist_o, rtol=1e-10, atol=0)
    assert np.abs(np.array(out["x"]) - x_o).max() <= 1e-10 * np.abs(x_o).max()
    assert [tuple(e) for e in out["events"]] == ev_o
    assert out["tot"] < 0.05 and out["mx"] < 0.01
    for k, i, bit in ev_o:  # rank 0 printed each event once, with the reference's text
        assert stdout.count("at index %d\n" % i) == 1


def test_sed_fatal_exits_with_status_1():
    out, stdout = run_worker("sed", (99, 3))
    assert out["exit"] == 1
    assert [tuple(e) for e in out["events"]] == [(1, 99, 0)]
    assert "[ECC] error detected at index 99\n" in stdout
