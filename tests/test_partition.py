"""The multi-rank decomposition on the CPU: the partition planner the C++ host uses
(host/partition.cpp, through libabft_host.so) checked property by property, and a whole
row-/column-partitioned CG run with world sizes 2 and 3 over torch.distributed (gloo), numpy
standing in for the GPU kernels -- same iteration count and residual history (1e-10) as one
process.  (The real kernels run the same shard geometry on the one GPU in test_gpu_cli.py.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from _oracle import COO, CSR, OracleMatrix, laplace5, random_spd, rhs

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
WORKER = os.path.join(HERE, "_partition_worker.py")


def lib():
    import ctypes as C
    return C.CDLL(os.path.join(ROOT, "abft_sparse_cg_amd", "libabft_host.so"))


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("matrix", ["laplace", "random"])
def test_planner_properties(fmt, world, matrix):
    from _partition_worker import local_spmv, plan
    cols, rows, vals, n = laplace5(40, 33) if matrix == "laplace" else random_spd(900, 9, seed=4)
    L = lib()
    plans = [plan(L, fmt, cols, rows, vals, n, world, me) for me in range(world)]
    out, gat = (rows, cols) if fmt == 0 else (cols, rows)
    b = plans[0]["bounds"]
    assert b[0] == 0 and b[-1] == n and all(b[g + 1] > b[g] for g in range(world))
    assert all(np.array_equal(p["bounds"], b) for p in plans)
    slot = max(b[g + 1] - b[g] for g in range(world))
    seen = np.zeros(len(vals), int)
    x = rhs(n, 3) - 0.5
    want = OracleMatrix(CSR if fmt == 0 else COO, "none", cols, rows, vals, n).spmv(x)
    xpad = np.zeros(slot * world)
    for g in range(world):
        xpad[g * slot:g * slot + b[g + 1] - b[g]] = x[b[g]:b[g + 1]]
    counts = []
    for me, p in enumerate(plans):
        assert p["slot"] == slot and p["n_pad"] == slot * world and p["out0"] == b[me] and p["n_loc"] == b[me + 1] - b[me]
        g = p["gidx"]
        assert np.all(np.diff(g.astype(np.int64)) > 0)  # ascending: the caller's order is kept
        if fmt == 0:
            assert np.array_equal(g, p["first"] + np.arange(p["nnz"]))  # CSR: one contiguous run
        seen[g] += 1
        assert np.array_equal(p["lout"], out[g] - b[me])
        owner = np.searchsorted(b, gat[g], side="right") - 1
        assert np.array_equal(p["pin"], owner * slot + (gat[g] - b[owner]))
        # the windows cover every remote read, the interior rows read nothing remote
        for k in np.nonzero(owner != me)[0]:
            lo, hi = p["need"][owner[k]]
            assert lo <= gat[g][k] - b[owner[k]] < hi
        lo, hi = p["interior"]
        inside = (p["lout"] >= lo) & (p["lout"] < hi)
        assert np.all(owner[inside] == me)
        # the shard's SpMV on the padded vector = its rows of the whole product, bit for bit
        y = local_spmv(p, vals[g], xpad)
        assert np.array_equal(y.view(np.uint64), want[b[me]:b[me + 1]].view(np.uint64))
        counts.append(p["nnz"])
    assert np.all(seen == 1)
    if world > 1 and matrix == "random":
        assert max(counts) < 1.3 * len(vals) / world  # cut by non-zeros, not by rows


def run_world(world, fmt, matrix):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), str(fmt), matrix],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            pytest.fail("rank job timed out")
    assert [p.returncode for p in procs] == [0] * world, outs[0][1][-2000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("matrix", ["laplace", "random"])
def test_partitioned_cg_over_gloo_matches_one_process(fmt, world, matrix):
    cols, rows, vals, n = laplace5(40, 33) if matrix == "laplace" else random_spd(900, 9, seed=4)
    o = OracleMatrix(CSR if fmt == 0 else COO, "none", cols, rows, vals, n)
    it, hist, x, fatal = o.cg(rhs(n, 1))
    out = run_world(world, fmt, matrix)
    assert out["it"] == it
    assert np.allclose(out["hist"], hist, rtol=1e-10, atol=0)
    assert np.abs(np.array(out["x"]) - x).max() <= 1e-10 * np.abs(x).max()
    assert out["windows"] == (matrix == "laplace")  # banded: halo windows; scattered: all-gather
    if matrix == "laplace":
        lo, hi = out["interior"]
        assert hi - lo > (out["bounds"][1] - out["bounds"][0]) // 2


def test_planner_awkward_shapes():
    """few rows per rank, all non-zeros in one row, empty rows at the cuts, as many ranks as rows"""
    from _partition_worker import plan
    L = lib()
    n = 7
    rows = np.array([3] * 7 + [6], np.uint32)          # everything in row 3, one element in the last row
    cols = np.array([0, 1, 2, 3, 4, 5, 6, 6], np.uint32)
    vals = np.arange(1.0, 9.0)
    for fmt in (0, 1):
        for world in (1, 2, 4, 7):
            plans = [plan(L, fmt, cols, rows, vals, n, world, me) for me in range(world)]
            b = plans[0]["bounds"]
            assert b[0] == 0 and b[-1] == n and all(b[g + 1] > b[g] for g in range(world))  # one output each at least
            seen = np.zeros(len(vals), int)
            for p in plans:
                seen[p["gidx"]] += 1
            assert np.all(seen == 1)
    import ctypes as C
    u32p = C.POINTER(C.c_uint32)
    rc = L.abft_plan_shard(0, cols.ctypes.data_as(u32p), rows.ctypes.data_as(u32p), vals.ctypes.data_as(C.POINTER(C.c_double)),
                           C.c_longlong(8), 7, 8, 0, None, None, None, None, None, None, C.c_longlong(0))
    assert rc == 1  # more ranks than rows: refused (the backend exits with a message)


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_outbox_layout_of_the_window_exchange(fmt, world):
    """host/partition.cpp plan_outboxes: how the halo windows are laid out in the senders' outboxes for
    the exchange through shared host memory (abft_hip_peer_exchange_*).  Every rank computes it alone
    from the windows all ranks know; here: both ends of every window agree, windows (+ their check
    word) do not overlap inside an outbox, all fit the common outbox size, and a host-side replay of
    the exchange -- copy out, copy in -- delivers exactly the entries every rank's SpMV reads."""
    import ctypes as C
    from _partition_worker import plan
    cols, rows, vals, n = laplace5(31, 47)
    L = lib()
    L.abft_plan_outboxes.restype = C.c_longlong
    plans = [plan(L, fmt, cols, rows, vals, n, world, me) for me in range(world)]
    slot = plans[0]["slot"]
    need = np.zeros((world, world, 2), dtype=np.int32)  # [reader][owner] = window of the owner's slot
    for q, p in enumerate(plans):
        for g in range(world):
            need[q, g] = p["need"][g] if g != q else (0, 0)
    flat = np.ascontiguousarray(need.reshape(-1))
    layouts, boxes = [], set()
    for me in range(world):
        out = np.zeros(4 * 64, dtype=np.int64)
        inn = np.zeros(4 * 64, dtype=np.int64)
        no, ni = C.c_int(), C.c_int()
        box = L.abft_plan_outboxes(flat.ctypes.data_as(C.POINTER(C.c_int)), world, slot, me,
                                   out.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(no),
                                   inn.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(ni), 64)
        boxes.add(box)
        layouts.append((out[:4 * no.value].reshape(-1, 4), inn[:4 * ni.value].reshape(-1, 4)))
    assert len(boxes) == 1 and min(boxes) > 0  # the same outbox size on every rank
    box = boxes.pop()
    for me, (out, inn) in enumerate(layouts):
        spans = sorted((int(o[3]), int(o[3]) + 8 * (int(o[2]) + 1)) for o in out)
        assert all(a1 <= b0 for (_, a1), (b0, _) in zip(spans, spans[1:])) and (not spans or spans[-1][1] <= box)
        assert all(int(o[3]) % 256 == 0 and 0 <= o[0] < world and o[0] != me for o in out)
        for peer, voff, count, boff in inn:  # the sender lists the same window, for me, at the same place
            theirs = layouts[peer][0]
            assert any(t[0] == me and t[1] == voff and t[2] == count and t[3] == boff for t in theirs)
            assert voff == peer * slot + need[me, peer, 0] and count == need[me, peer, 1] - need[me, peer, 0]
        assert len(inn) == sum(1 for g in range(world) if g != me and need[me, g, 1] > need[me, g, 0])
    # replay on the host: every rank's gathered vector ends up holding what its shard reads
    x = rhs(n, 7)
    b = plans[0]["bounds"]
    full = [np.full(slot * world, np.nan) for _ in range(world)]
    for g in range(world):
        full[g][g * slot:g * slot + b[g + 1] - b[g]] = x[b[g]:b[g + 1]]
    outboxes = [np.zeros(box // 8) for _ in range(world)]
    for me, (out, _) in enumerate(layouts):
        for peer, voff, count, boff in out:
            outboxes[me][boff // 8:boff // 8 + count] = full[me][voff:voff + count]
    for me, (_, inn) in enumerate(layouts):
        for peer, voff, count, boff in inn:
            full[me][voff:voff + count] = outboxes[peer][boff // 8:boff // 8 + count]
    for me, p in enumerate(plans):
        got = full[me][p["pin"]]
        owner = p["pin"] // slot
        assert not np.any(np.isnan(got))
        assert np.array_equal(got, x[np.asarray(b)[owner] + p["pin"] % slot])
