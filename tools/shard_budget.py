#!/usr/bin/env python3
"""Per-rank kernel time of the row-partitioned CG iteration, measured on ONE GPU, and what it
projects to for 2 / 4 / 8 GPUs (no multi-GPU box is available to this repository's builder:
everything below the line "PROJECTED" is arithmetic on stated assumptions, not a measurement).

    python tools/shard_budget.py [--spec random:4194304,24,1] [--mode secded] [--ranks 1,2,4,8]

For each world size G and a few ranks k it builds rank k's shard exactly as the C++ host does
(row block cut by non-zeros, columns re-based to the slot-padded gathered vector:
host/partition.cpp), and times the iteration's kernels through the C ABI with the scalars
resident on the device and the iteration replayed as a hipGraph (what host/cg-csr --bench
runs, minus the collectives):  spmv+dot, calc_xr (r half), calc_p (+x half).
Measured besides: the host cost of one graph launch.  For G > 1 the replayed iteration also holds
the two scalar all-reduces as the multi-process host runs them -- over the peer board, in the tails
of the kernels that finish the shard's sums (abft_hip_peer_board_fuse), on a board of ONE rank: two
PCIe crossings each, i.e. what they cost when all peers arrive together; the skew between real
ranks is not in it.

Projection per iteration at G ranks:
    T(G) = max_k kernels(G, k)  [incl. the two board all-reduces]  + T_exchange(G)
    T_exchange = all-gather of 8 * slot bytes per rank over xGMI: each rank receives (G-1) slots,
                 one from each peer over its own link, concurrently: 8 * slot / (LINK_GBPS * EFF)
                 (banded matrices exchange halo windows of a few KB through shared host memory
                 instead, one kernel: T_WINDOW_US = 9 (device memory; 14 through host memory), measured between two streams of one GPU
                 with windows of config 2's size, tools/peer_latency.py)
with LINK_GBPS = 153 (MI355X_MICROARCH.md: 7 links x ~153 GB/s), EFF = 0.7.
Prints a markdown table (commit it under profiles/)."""
import argparse
import ctypes as C
import json
import mmap
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# T_WINDOW_US: measured between two contexts on ONE device (a lower bound for two devices: no xGMI hop in it);
# 14 us through shared host memory (round 2).  Both bounds are printed.
LINK_GBPS, EFF, T_WINDOW_US, T_WINDOW_HOST_US = 153.0, 0.7, 9.0, 14.0


def shard(spec, G, k):
    from abft_sparse_cg_amd import generators
    n = generators.dim(spec)
    bounds = generators.partition(spec, G)
    slot = max(bounds[g + 1] - bounds[g] for g in range(G))
    cols, rows, vals, _ = generators.generate(spec, bounds[k], bounds[k + 1])
    b = np.asarray(bounds)
    owner = np.searchsorted(b, cols, side="right") - 1
    pin = (owner * slot + (cols - b[owner])).astype(np.uint32)
    remote = owner != k
    window = 0
    for g in range(G):
        m = owner == g
        if g != k and m.any():
            off = cols[m] - b[g]
            window += int(off.max() - off.min() + 1)
    return pin, (rows - bounds[k]).astype(np.uint32), vals, bounds[k + 1] - bounds[k], slot * G, slot, window, \
        int(remote.sum()), n


def time_shard(mode, pin, lrows, vals, n_loc, n_pad, slot, k, iters=60, allreduce=False):
    import abft_sparse_cg_amd as amd
    from abft_sparse_cg_amd import capi
    ctx = amd.HIPContext(mode, "csr", device=0)
    L, h = ctx.L, ctx.h
    A = ctx.create_matrix(pin, lrows, vals, n_loc, len(vals), n_in=n_pad, index_base=0)
    layout, launches = ctx.matrix_info(A)
    pfull = ctx.create_vector(n_pad)
    ctx.upload(pfull, np.random.default_rng(1).random(n_pad))
    p = ctx.view_vector(pfull, k * slot, n_loc)
    x, r, w = (ctx.create_vector(n_loc) for _ in range(3))
    for v in (x, r, w):
        ctx.upload(v, np.random.default_rng(2).random(n_loc))
    sc = ctx.create_vector(6)
    ctx.upload(sc, np.array([1.0, 0.0, 1.0, 0.0, 1.0, 0.0]))
    base = sc.device_ptr

    board = None
    if allreduce:  # a board of one rank: the all-reduce kernel's own cost
        # the board as the multi-process host sets it up first (round 3): a copy in device memory
        # (ABFT_SHARD_BUDGET_BOARD=host: the shared-host-memory board of round 2)
        if os.environ.get("ABFT_SHARD_BUDGET_BOARD") == "host":
            board = mmap.mmap(-1, L.abft_hip_peer_board_bytes())
            capi.check(L.abft_hip_peer_board_attach(h, C.addressof(C.c_char.from_buffer(board)), len(board), 0, 1, 5.0))
        else:
            bp = C.c_void_p()
            capi.check(L.abft_hip_peer_board_device_alloc(h, C.byref(bp)))
            boards = (C.c_void_p * 1)(bp.value)
            capi.check(L.abft_hip_peer_board_attach_device(h, boards, 0, 1, 5.0))
        capi.check(L.abft_hip_peer_board_fuse(h, 1))  # in the tails of the reductions, as host/HIPContext.cpp runs them

    three = os.environ.get("ABFT_CG_TAIL") == "0"  # rounds 2-3: fold, calc_r, calc_px as three launches

    def it(parity):
        cur, nxt, pw = base + 16 * parity, base + 16 * (1 - parity), base + 32
        if not three:  # as host/HIPContext.cpp runs it since round 4: everything behind the SpMV in one launch
            capi.check(L.abft_hip_cg_iteration_dev(h, A.h, pfull.h, k * slot, capi.PART_ALL, x.h, r.h, p.h, w.h, cur, pw, nxt))
            return
        capi.check(L.abft_hip_spmv_dot_dev(h, A.h, pfull.h, w.h, k * slot, pw))
        capi.check(L.abft_hip_calc_xr_ratio_dev(h, x.h, r.h, p.h, w.h, cur, pw, nxt))
        capi.check(L.abft_hip_calc_p_ratio_dev(h, p.h, r.h, nxt, cur))
    it(0); it(1)
    ctx.synchronize()
    graphs = []
    for parity in (0, 1):
        capi.check(L.abft_hip_graph_begin(h))
        it(parity)
        g = C.c_void_p()
        capi.check(L.abft_hip_graph_end(h, C.byref(g)))
        graphs.append(g)
    for _ in range(4):
        capi.check(L.abft_hip_graph_launch(graphs[0])); capi.check(L.abft_hip_graph_launch(graphs[1]))
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        capi.check(L.abft_hip_graph_launch(graphs[i & 1]))
    t_host = time.perf_counter() - t0
    ctx.synchronize()
    t_all = time.perf_counter() - t0
    # SpMV alone, bracketed
    ctx.profile(1 << capi.K_SPMV, stride=1)
    for i in range(20):
        capi.check(L.abft_hip_spmv_dot_dev(h, A.h, pfull.h, w.h, k * slot, base + 32))
    ms, cnt = ctx.profile_read(capi.K_SPMV)
    for g in graphs:
        L.abft_hip_graph_destroy(g)
    ctx.close()
    return {"iter_us": t_all / iters * 1e6, "host_launch_us": t_host / iters * 1e6, "spmv_us": ms * 1e3 / max(cnt, 1),
            "layout": layout, "launches": launches}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spec", default="random:4194304,24,1")
    ap.add_argument("--mode", default="secded")
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    rows = []
    for G in [int(v) for v in a.ranks.split(",")]:
        for k in sorted({0, G // 2, G - 1}):
            pin, lrows, vals, n_loc, n_pad, slot, window, remote, n = shard(a.spec, G, k)
            t = time_shard(a.mode, pin, lrows, vals, n_loc, n_pad, slot, k, allreduce=G > 1)
            use_windows = G > 1 and window * 2 * G < n * (G - 1)  # this rank's share of the planner's rule
            t.update(G=G, rank=k, rows=int(n_loc), nnz=int(len(vals)), slot=int(slot), window_entries=window,
                     exchange="windows" if use_windows else ("all-gather" if G > 1 else "none"))
            rows.append(t)
            print("measured G=%d rank %d: %d rows, %d nnz, layout %s: iteration kernels %.1f us (SpMV %.1f us, graph "
                  "launch %.1f us of host time)" % (G, k, n_loc, len(vals), t["layout"], t["iter_us"], t["spmv_us"],
                                                    t["host_launch_us"]), flush=True)
    print("\nPROJECTED (assumptions in the module docstring; no multi-GPU run behind it)\n")
    print("Assumptions, all of them unmeasured between two devices: the all-gather moves 8 * slot bytes per rank over each of "
          "the (G-1) links at %.0f GB/s x %.1f, fully exposed; a halo-window exchange costs %.0f us (device memory, measured "
          "between two contexts on ONE device: a lower bound, no xGMI hop in it; %.0f us through shared host memory: the "
          "second column of speed-ups); the two board all-reduces are in the measured kernels, on a board of one rank (no "
          "skew between ranks, no link latency).\n" % (LINK_GBPS, EFF, T_WINDOW_US, T_WINDOW_HOST_US))
    print("| GPUs | slowest rank's kernels incl. the two board all-reduces, us (measured, 1 GPU) | of which SpMV | exchange, us "
          "(assumed) | iteration, us | speed-up vs 1 GPU | ... with the host-memory window bound |")
    print("|---|---|---|---|---|---|---|")
    t1 = t1h = None
    out = []
    for G in sorted({r["G"] for r in rows}):
        rs = [r for r in rows if r["G"] == G]
        slowest = max(rs, key=lambda r: r["iter_us"])
        kern = slowest["iter_us"]
        exh = None
        if G == 1:
            ex = 0.0
        elif rs[0]["exchange"] == "windows":
            ex, exh = T_WINDOW_US, T_WINDOW_HOST_US
        else:
            ex = 8.0 * rs[0]["slot"] / (LINK_GBPS * EFF * 1e3)
        tot = kern + ex
        toth = kern + (exh if exh is not None else ex)
        if G == 1:
            t1, t1h = tot, toth
        out.append({"G": G, "kernels_us": kern, "spmv_us": slowest["spmv_us"], "exchange_us": ex, "iteration_us": tot,
                    "speedup": (t1 / tot) if t1 else None, "speedup_host_window_bound": (t1h / toth) if t1h else None})
        print("| %d | %.1f | %.1f | %.1f | %.1f | %s | %s |" % (G, kern, slowest["spmv_us"], ex, tot,
                                                          ("%.2fx" % (t1 / tot)) if t1 else "-",
                                                          ("%.2fx" % (t1h / toth)) if t1h else "-"))
    if a.json:
        json.dump({"spec": a.spec, "mode": a.mode, "measured": rows, "projected": out,
                   "assumptions": {"link_GBps": LINK_GBPS, "efficiency": EFF, "window_exchange_us": T_WINDOW_US,
                                   "allreduce": "measured: the peer-board kernel on a board of one rank"}},
                  open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
