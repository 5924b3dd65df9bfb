"""The sharded solver with the REAL engine (HipEngine -> libabft_hip.so) on the
one GPU of the test box: world_size 1 over the nccl (RCCL) backend, which
exercises the device-memory aliasing, the stream hand-over and the collective
calls end to end.  Multi-rank logic is covered on CPU by test_distributed_gloo."""
import os
import socket

import numpy as np
import pytest

from _oracle import CSR, OracleMatrix, laplace5, rhs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,flip", [("none", None), ("secded", (777, [40])), ("sec7", (5, [70]))])
def test_sharded_engine_world1_matches_oracle(pg, mode, flip):
    from abft_sparse_cg_amd.distributed import HipEngine, ShardedCG
    cols, rows, vals, n = laplace5(48, 48)
    b = rhs(n, 1)
    o = OracleMatrix(CSR, mode, cols, rows, vals, n)
    if flip:
        o.inject(*flip)
    it_o, hist_o, x_o, _ = o.cg(b)
    ev_o, _ = o.events()
    eng = HipEngine(mode, "csr", device=0)
    try:
        cg = ShardedCG(eng, cols, rows, vals, [0, n], 0, mode)
        cg.set_rhs(b)
        if flip:
            eng.inject(cg.A, *flip)
        hist = []
        it, rr = cg.solve(on_iteration=lambda i, r: hist.append(r))
        assert it == it_o
        assert np.allclose(hist, hist_o, rtol=1e-10, atol=0)
        x = cg.gather_x()
        assert np.abs(x - x_o).max() <= 1e-10 * np.abs(x_o).max()
        assert cg.events == ev_o
        tot, mx = cg.residual_check()
        assert tot < 0.05 and mx < 0.01
    finally:
        eng.close()


def test_sed_fatal_exits_with_status_1(pg, capfd):
    from abft_sparse_cg_amd.distributed import HipEngine, ShardedCG
    cols, rows, vals, n = laplace5(32, 32)
    eng = HipEngine("sed", "csr", device=0)
    try:
        cg = ShardedCG(eng, cols, rows, vals, [0, n], 0, "sed")
        cg.set_rhs(rhs(n, 1))
        eng.inject(cg.A, 99, [3])
        with pytest.raises(SystemExit) as e:
            cg.solve()
        assert e.value.code == 1
        assert "[ECC] error detected at index 99\n" in capfd.readouterr().out
    finally:
        eng.close()
