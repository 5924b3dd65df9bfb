"""One rank of the CPU test of the multi-rank decomposition (tests/test_partition.py): the
product's partition planner (host/partition.cpp through libabft_host.so) + numpy in place of
the GPU kernels + torch.distributed (gloo) in place of RCCL.  Runs the CG loop of cg.cpp:87-118
on this rank's shard: exchange of the search vector (halo windows or all-gather, as the planner
decides), SpMV in the caller's element order, two all-reduces per iteration.

    python _partition_worker.py RANK WORLD PORT FMT MATRIX
prints one line "RESULT {json}" (rank 0)."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def plan(L, fmt, cols, rows, vals, n, world, me):
    nnz = len(vals)
    bounds = np.zeros(world + 1, np.int32)
    sc = np.zeros(8, np.int64)
    need = np.zeros(2 * world, np.int32)
    lout, pin, gidx = (np.zeros(max(nnz, 1), np.uint32) for _ in range(3))
    u32p = C.POINTER(C.c_uint32)
    rc = L.abft_plan_shard(fmt, cols.ctypes.data_as(u32p), rows.ctypes.data_as(u32p),
                           vals.ctypes.data_as(C.POINTER(C.c_double)), C.c_longlong(nnz), n, world, me,
                           bounds.ctypes.data_as(C.POINTER(C.c_int)), sc.ctypes.data_as(C.POINTER(C.c_longlong)),
                           need.ctypes.data_as(C.POINTER(C.c_int)), lout.ctypes.data_as(u32p), pin.ctypes.data_as(u32p),
                           gidx.ctypes.data_as(u32p), C.c_longlong(nnz))
    assert rc == 0
    k = int(sc[4])
    return dict(bounds=bounds, slot=int(sc[0]), n_pad=int(sc[1]), out0=int(sc[2]), n_loc=int(sc[3]), nnz=k,
                first=int(sc[5]), interior=(int(sc[6]), int(sc[7])), need=need.reshape(world, 2), lout=lout[:k],
                pin=pin[:k], gidx=gidx[:k])


def local_spmv(p, vals_loc, xpad):
    """outputs summed in the caller's element order, products formed separately (no FMA)"""
    prod = vals_loc * np.where(p["pin"] < p["n_pad"], xpad[np.minimum(p["pin"], p["n_pad"] - 1)], 0.0)
    y = np.zeros(p["n_loc"])
    # np.add.at adds in index order: the caller's order within each output
    np.add.at(y, p["lout"], prod)
    return y


def main():
    rank, world, port, fmt, matrix = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import torch
    import torch.distributed as dist
    from _oracle import laplace5, random_spd, rhs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cols, rows, vals, n = laplace5(40, 33) if matrix == "laplace" else random_spd(900, 9, seed=4)
    L = C.CDLL(os.path.join(ROOT, "abft_sparse_cg_amd", "libabft_host.so"))
    p = plan(L, fmt, cols, rows, vals, n, world, rank)
    vals_loc = vals[p["gidx"]]
    needs = [torch.zeros(world, 2, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(needs, torch.from_numpy(p["need"].astype(np.int32)))
    needs = [t.numpy() for t in needs]  # needs[src][dst] = window of dst's slot that src reads
    moved = sum(int(w[1] - w[0]) for nd in needs for w in nd)
    windows = moved * 2 < n * (world - 1)
    slot, me0 = p["slot"], rank * p["slot"]

    def exchange(v_loc):
        xpad = np.zeros(p["n_pad"])
        xpad[me0:me0 + p["n_loc"]] = v_loc
        if not windows:
            mine = np.zeros(slot)
            mine[:p["n_loc"]] = v_loc
            parts = [torch.zeros(slot, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(mine))
            return np.concatenate([t.numpy() for t in parts])
        reqs, bufs = [], []
        for g in range(world):
            if g == rank:
                continue
            lo, hi = needs[g][rank]  # what g reads of my slot
            if hi > lo:
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(v_loc[lo:hi])), g))
            lo, hi = needs[rank][g]
            if hi > lo:
                t = torch.zeros(hi - lo, dtype=torch.float64)
                bufs.append((g, lo, hi, t))
                reqs.append(dist.irecv(t, g))
        for r in reqs:
            r.wait()
        for g, lo, hi, t in bufs:
            xpad[g * slot + lo:g * slot + hi] = t.numpy()
        return xpad

    def allsum(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t[0])

    b = rhs(n, 1)[p["out0"]:p["out0"] + p["n_loc"]]
    x = np.zeros(p["n_loc"])
    r = b.copy()
    pv = r.copy()
    rr = allsum(float(r @ r))
    hist = []
    it = 0
    while it < 1000 and rr > 1e-3:
        w = local_spmv(p, vals_loc, exchange(pv))
        alpha = rr / allsum(float(pv @ w))
        x += alpha * pv
        r -= alpha * w
        rr_new = allsum(float(r @ r))
        pv = r + (rr_new / rr) * pv
        rr = rr_new
        hist.append(rr)
        it += 1
    xs = [None] * world
    dist.all_gather_object(xs, x.tolist())
    if rank == 0:
        print("RESULT " + json.dumps({"it": it, "hist": hist, "x": sum(xs, []), "windows": bool(windows),
                                      "interior": p["interior"], "bounds": p["bounds"].tolist()}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
