#!/usr/bin/env python3
"""bench.py -- CG iterations/s and SpMV effective HBM GB/s of the hip target.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one CG iteration driven through the C ABI exactly as the reference
driver drives a backend (cg.cpp:97-112): spmv, dot, calc_xr, calc_p, with the
two scalars returned to the host.  The workload is BASELINE.json configs[1]:
`cg-csr -t hip -m none` on the synthetic 5-point Laplacian 3162 x 3162
(N = 9,998,244, nnz = 49,978,572), b = uniform(0,1), x0 = 0, fixed iteration
count (-c 0).  For N > 1 the same matrix is row-partitioned over the ranks
(strong scaling): every rank of the launcher starts the C++ driver (host/cg-csr
--bench W,K,5: one process per GPU, HIPContext.cpp + comm*.cpp); each builds its
own shard, the search-vector exchange is an RCCL all-gather (halo windows of a
banded matrix: pushed through IPC-mapped device memory), the two scalar
all-reduces go over the peer board or RCCL (what each rank used is recorded in
`transport_by_rank`).  torch.distributed is only the launcher, plus one gloo
all-reduce of the ranks' exit codes.

Rank 0 prints ONE JSON line; see the keys at the bottom.  The matrix, vectors and
the CPU baseline are built outside the timed region; inputs are resident in HBM
when the clock starts.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy peak reported beside it
BLOCKS = 5              # N = 1: the K timed steps are repeated this many times; the median block is reported
CPU_RUNS = 5            # CPU baseline: runs per figure (reference run_benchmark:3,19-25: mean (min / max))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spec", default="laplace5:3162,3162", help="synthetic matrix (generators.cpp)")
    ap.add_argument("--mode", default="none")
    ap.add_argument("--fmt", default="csr", choices=["csr", "coo"])
    ap.add_argument("--cpu-iters", type=int, default=160,
                    help="CG iterations of the CPU baseline sample on all host cores, in 5 runs (mean / min / max), "
                         "~10 s (0 = skip); a quarter of that is also timed on one core")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket SpMV launches with HIP events")
    ap.add_argument("--profile-all", action="store_true",
                    help="bracket all four kernels, not only the SpMV (costs ~10%% of the iteration rate)")
    ap.add_argument("--no-probe", dest="probe", action="store_false",
                    help="skip the streaming-copy bandwidth probe (the measured-peak denominator beside 8 TB/s)")
    ap.add_argument("--no-extras", dest="extras", action="store_false",
                    help="N = 1: skip the short legs on the other BASELINE.json configurations")
    return ap.parse_args()


# The other single-GPU configurations of BASELINE.json (north_star's SpMV+ECC target is quoted
# on the first two), each run as a short leg AFTER the headline's timed loop:
# (key, spec, fmt, mode, BASELINE.json config it stands for)
EXTRA_LEGS = [
    ("config2_sed", "laplace5:3162,3162", "csr", "sed", "configs[2] matrix, -m sed, fault-free (detection itself: tests)"),
    ("config2_secded", "laplace5:3162,3162", "csr", "secded", "configs[1] matrix, -m secded"),
    ("config4_shard1", "random:4194304,24,1", "csr", "secded", "configs[3] matrix (100 M nnz) on ONE GPU"),
    ("config5", "powerlaw:2097152,2", "coo", "sec7", "configs[4]: cg-coo -m sec (= sec7)"),
]


def spmv_bytes(fmt, n, nnz):
    """Algorithmic bytes of one SpMV (SURVEY 8d / BASELINE.md section 5)."""
    return 12 * nnz + 20 * n + 4 if fmt == "csr" else 16 * nnz + 16 * n


def traffic_from_profile(workload):
    """HBM bytes per SpMV launch from the committed rocprofv3 PMC summary, if any."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        d = json.load(open(path))
        return d.get(workload, {}).get("spmv_hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def extra_leg(spec, fmt, mode, triplets=None, iters=32, warm=4):
    """One short leg: `iters` CG iterations with EVERY SpMV bracketed by HIP events on the
    context's stream.  -> dict for the bench line."""
    import numpy as np

    import abft_sparse_cg_amd as amd
    from abft_sparse_cg_amd import capi, generators
    from abft_sparse_cg_amd.context import fdiv

    cols, rows, vals, n = triplets if triplets else generators.generate(spec)
    nnz = len(vals)
    ctx = amd.HIPContext(mode, fmt, device=0)
    A = ctx.create_matrix(cols, rows, vals, n, nnz)
    layout, launches = ctx.matrix_info(A)
    b, x, r, p, w = (ctx.create_vector(n) for _ in range(5))
    ctx.upload(b, generators.reference_rhs(n))
    ctx.upload(x, np.zeros(n))
    ctx.copy_vector(r, b)
    ctx.copy_vector(p, r)
    rr = ctx.dot(r, r)
    t0 = None
    for it in range(warm + iters):
        if it == warm:
            ctx.profile(1 << capi.K_SPMV, stride=1)
            ctx.synchronize()
            t0 = time.perf_counter()
        ctx.spmv(A, p, w)
        alpha = fdiv(rr, ctx.dot(p, w))
        rr_new = ctx.calc_xr(x, r, p, w, alpha)
        ctx.calc_p(p, r, fdiv(rr_new, rr))
        rr = rr_new
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ms, cnt = ctx.profile_read(capi.K_SPMV)
    ctx.close()
    us = ms * 1e3 / max(cnt, 1)
    byts = spmv_bytes(fmt, n, nnz)
    ach = byts / us / 1e3
    return {"workload": "cg-%s -t hip -m %s, synthetic %s" % (fmt, mode, spec), "N": n, "nnz": nnz, "layout": layout,
            "sampled_spmvs": cnt, "avg_spmv_us": round(us, 2), "launches_per_spmv": launches,
            "avg_launch_us": round(us / launches, 2), "algorithmic_bytes_per_spmv": byts,
            "achieved": round(ach, 1), "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
            "traffic": traffic_from_profile("%s/%s/%s" % (spec, fmt, mode)),
            "it_per_s_with_brackets": round(iters / dt, 1)}


def single(args):
    import numpy as np
    import torch

    import abft_sparse_cg_amd as amd
    from abft_sparse_cg_amd import capi, generators
    from abft_sparse_cg_amd.context import fdiv

    cols, rows, vals, n = generators.generate(args.spec)
    nnz = len(vals)
    ctx = amd.HIPContext(args.mode, args.fmt, device=0)
    A = ctx.create_matrix(cols, rows, vals, n, nnz)
    b, x, r, p, w = (ctx.create_vector(n) for _ in range(5))
    ctx.upload(b, generators.reference_rhs(n))  # glibc rand(), default seed: the reference's b (cg.cpp:66-74)
    ctx.upload(x, np.zeros(n))
    ctx.copy_vector(r, b)
    ctx.copy_vector(p, r)
    state = {"rr": ctx.dot(r, r)}

    def step():
        ctx.spmv(A, p, w)
        pw = ctx.dot(p, w)
        alpha = fdiv(state["rr"], pw)
        rr_new = ctx.calc_xr(x, r, p, w, alpha)
        ctx.calc_p(p, r, fdiv(rr_new, state["rr"]))
        state["rr"] = rr_new

    def restart():
        # r = b; p = r; rr = r.r (cg.cpp:87-91), then the warm-up steps
        ctx.copy_vector(r, b)
        ctx.copy_vector(p, r)
        state["rr"] = ctx.dot(r, r)
        for _ in range(args.warmup):
            step()

    restart()
    if not args.no_profile:
        # HIP-event brackets inside the timed region on sampled SpMV launches (a bracket
        # serialises the launches around it: sampling all of them costs ~3 %): every 4th,
        # or more often when the run is short, so that at least ~25 launches are averaged
        stride = 1 if args.profile_all else max(1, min(4, args.steps // 25))
        ctx.profile(0xF if args.profile_all else 1 << capi.K_SPMV, stride=stride)
    # The timed region: exactly `steps` iterations between two synchronisations -- BLOCKS times (a 5 ms window
    # is shorter than this pool's run-to-run spread), every block the whole run again (r = b, warm-up, timed
    # steps: rr at the end is rr after warmup + steps iterations, whatever BLOCKS is): `value` comes from the
    # median block, the fastest and the slowest ride along as value_max / value_min.
    block_dt = []
    for blk in range(BLOCKS):
        if blk:
            restart()
        ctx.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ctx.synchronize()
        torch.cuda.synchronize()
        block_dt.append(time.perf_counter() - t0)
    dt = sorted(block_dt)[len(block_dt) // 2]

    roof = None
    kernels = {}
    if not args.no_profile:
        names = {capi.K_SPMV: "spmv", capi.K_DOT: "dot", capi.K_CALC_XR: "calc_xr", capi.K_CALC_P: "calc_p"}
        byts = {"spmv": spmv_bytes(args.fmt, n, nnz), "dot": 16 * n, "calc_xr": 48 * n, "calc_p": 24 * n}
        for k, name in names.items():
            ms, cnt = ctx.profile_read(k)
            if cnt:
                us = ms * 1e3 / cnt
                kernels[name] = {"avg_us": round(us, 2), "launches": cnt, "GBps": round(byts[name] / us / 1e3, 1)}
                if name in ("calc_xr", "calc_p") and os.environ.get("ABFT_HIP_FUSE_X", "1") != "0":
                    # GBps is on the reference's byte count; the x update actually runs inside calc_p
                    kernels[name]["note"] = "x += alpha p deferred from calc_xr into calc_p (moves 24N / 40N bytes)"
                if name == "dot" and os.environ.get("ABFT_HIP_FUSE_DOT", "1") != "0":
                    # dot(p,w) is formed inside the SpMV; what is timed here is the one-block fold of its partials
                    kernels[name] = {"avg_us": round(us, 2), "launches": cnt, "note": "fold of the SpMV's fused p.w partials"}
        if "spmv" in kernels:
            ach = kernels["spmv"]["GBps"]
            workload = "%s/%s/%s" % (args.spec, args.fmt, args.mode)
            roof = {"bound": "hbm", "kernel": "spmv_%s_kernel<%s>" % (args.fmt, args.mode), "achieved": ach,
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                    "traffic": traffic_from_profile(workload),
                    # (traffic is NOT measured by this run: HBM bytes per launch from separate rocprofv3 --pmc passes)
                    "traffic_source": "profiles/pmc_summary.json (committed; profiles/collect_r04.sh regenerates it)",
                    "avg_launch_us": kernels["spmv"]["avg_us"], "algorithmic_bytes_per_launch": byts["spmv"]}
    probe = None
    if args.probe:
        c, rd = ctx.stream_probe(1 << 30, 10)
        probe = {"copy_GBps": round(c, 1), "read_GBps": round(rd, 1)}
        if roof:
            # a measured denominator beside the 8 TB/s spec: a streaming read of 1 GiB on this box (the
            # SpMV is ~90 % reads; the copy probe's figure rides along in stream_probe, no fraction taken on it)
            roof["measured_read_peak"] = probe["read_GBps"]
            roof["frac_of_measured_read"] = round(roof["achieved"] / rd, 4)
    rr_final = state["rr"]
    # how the library ran the loop behind the unchanged API: iterations whose r and x / p halves it had run ahead of the
    # caller's calc_xr / calc_p (include/abft_hip.h: abft_hip_speculation_stats; ABFT_HIP_SPECULATE=0 turns that off)
    import ctypes
    taken, dropped = ctypes.c_long(), ctypes.c_long()
    capi.check(ctx.L.abft_hip_speculation_stats(ctx.h, ctypes.byref(taken), ctypes.byref(dropped)))
    spec_stats = {"taken_over": taken.value, "dropped": dropped.value}
    ctx.close()

    # ---- the other single-GPU configurations, outside the headline's timed loop ----
    extras = None
    if args.extras and not args.no_profile:
        extras = {}
        for key, spec, fmt, mode, what in EXTRA_LEGS:
            same = spec == args.spec
            try:
                leg = extra_leg(spec, fmt, mode, (cols, rows, vals, n) if same else None)
                leg["stands_for"] = what
            except Exception as e:  # noqa: BLE001 -- a leg that cannot run is reported, not hidden
                leg = {"error": repr(e)[:300], "stands_for": what}
            extras[key] = leg
        # The loop the N > 1 lines run (host/cg-csr --bench: scalars device-resident, the iteration replayed as a
        # hipGraph), by one process on this GPU: the like-for-like N = 1 point of the scaling curve (`value` above
        # is the reference driver's loop, two scalars back to the host per iteration, ~9 % slower).
        graph_legs = [("config2_graph_loop", args.spec, args.fmt, args.mode,
                       "N = 1 with the fixed-iteration loop that bench.py --gpus N > 1 times as `value`"),
                      ("config4_graph_loop", os.environ.get("ABFT_BENCH_EXTRA_SPEC", "random:4194304,24,1"), "csr", "secded",
                       "N = 1 base of extra_legs.config4 of the --gpus N > 1 lines (BASELINE.json configs[3], the matrix "
                       "north_star's 6x target is quoted on): same loop, one process")]
        for key, spec, fmt, mode, what in graph_legs:
            try:
                job = cpp_job(args, spec, mode, fmt, False)
                leg = {"workload": "cg-%s -t hip -m %s --bench %d,%d,%d, synthetic %s, one process"
                                   % (fmt, mode, args.warmup, args.steps, BLOCKS, spec)}
                leg.update(block_stats(job, args.steps))
                leg.update({"N": job["N"], "nnz": job["nnz"], "graph_replay": job["graph_replay"], "stands_for": what})
                if "first_attempt" in job:
                    leg["first_attempt"] = job["first_attempt"]
                extras[key] = leg
            except SystemExit as e:
                extras[key] = {"error": str(e)[:300], "stands_for": what}

    cpu = None
    if args.cpu_iters > 0 and args.fmt == "csr":
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import baseline  # the CPU checker, timed as a baseline only
        # the reference's protocol (run_benchmark:3,19-25): NUM_RUNS = 5 runs, mean (min / max) -- here of
        # CG iterations/s over 5 runs of cpu_iters / 5 iterations each, on all host cores and on one
        per = max(args.cpu_iters // CPU_RUNS, 2)

        def stats(res):
            v = res["it_per_s_runs"]
            return {"value": round(sum(v) / len(v), 3), "min": round(min(v), 3), "max": round(max(v), 3), "runs": len(v),
                    "cores": res["cores"], "sample": "%d runs of %d CG iterations, %.1f s in all"
                                                     % (len(v), res["iters"], res["seconds"])}

        res = baseline.time_cg(cols, rows, vals, n, args.mode, per, runs=CPU_RUNS)
        one = baseline.time_cg(cols, rows, vals, n, args.mode, max(per // 4, 2), threads=1, runs=CPU_RUNS)
        cpu = stats(res)
        cpu.update({"unit": "CG iterations/s", "cpu_model": cpu_model(), "kind": res["kind"],
                    "sample": "mean (min / max) of %d runs of %d CG iterations each of the same matrix (%s, -m %s), "
                              "OpenMP spmv + serial vector ops as the reference, %.1f s in all"
                              % (CPU_RUNS, res["iters"], args.spec, args.mode, res["seconds"]),
                    "one_core": stats(one)})
        if args.extras and args.mode != "secded":
            cpu["secded"] = stats(baseline.time_cg(cols, rows, vals, n, "secded", max(per // 2, 2), runs=CPU_RUNS))
    return dt, n, nnz, roof, kernels, cpu, probe, rr_final, extras, block_dt, spec_stats


def agree_codes(code):
    """Every rank of the launcher learns every rank's exit status of the job just run (one all-gather over a
    gloo group on the launcher's own store: CPU only, nothing to do with the job's own transports), so that
    all ranks take the same decision about it.  One rank: [code]."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return [int(code)]
    import datetime

    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        # (one node: the loopback interface always works, a host name that does not resolve does not)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
    t = torch.zeros(world, dtype=torch.int64)
    t[int(os.environ.get("RANK", "0"))] = int(code)
    dist.all_reduce(t)
    return [int(v) for v in t]


def run_cpp(cmd, env):
    import subprocess
    return subprocess.run(cmd, capture_output=True, text=True, env=env)


def parse_cpp(p):
    """rank 0's stdout of `cg-* --bench` -> dict (other ranks: their stdout is discarded by the backend)"""
    import re
    out = {"stderr": p.stderr}
    m = re.search(r"^bench: ranks (\d+) warmup (\d+) steps (\d+) seconds ([0-9.]+) iterations_per_second ([0-9.]+) rr (\S+)$",
                  p.stdout, re.M)
    if m:
        out.update(ranks=int(m.group(1)), seconds=float(m.group(4)), rr=float.fromhex(m.group(6)))
        # the K timed steps ran as B back-to-back blocks; `seconds` above is the median block
        bl = re.search(r"^bench_blocks: blocks (\d+) iterations_per_block (\d+) seconds((?: [0-9.]+)+)$", p.stdout, re.M)
        if bl:
            out.update(block_seconds=[float(t) for t in bl.group(3).split()], iterations_per_block=int(bl.group(2)))
        h = re.search(r"^matrix size +=\s+(\d+) x", p.stdout, re.M)
        z = re.search(r"^number of non-zeros +=\s+(\d+) ", p.stdout, re.M)
        out.update(N=int(h.group(1)), nnz=int(z.group(1)))
        q = re.search(r"^bench_spmv: rank 0 spmvs (\d+) brackets (\d+) total_us ([0-9.]+) local_rows (\d+) local_nnz (\d+)$",
                      p.stdout, re.M)
        if q:
            out["spmv"] = {"spmvs": int(q.group(1)), "brackets": int(q.group(2)), "total_us": float(q.group(3)),
                           "rows": int(q.group(4)), "nnz": int(q.group(5))}
        # what each rank used for the two all-reduces and the exchange, and what RCCL itself says about
        # its communicator (HIPContext.cpp run_fixed): the record of an unattended multi-GPU run
        out["transport"] = re.findall(r"^bench_transport: (.*)$", p.stdout, re.M)
    return out


def cpp_job(args, spec, mode, fmt, profile):
    """One fixed-iteration run of the C++ driver (host/cg-csr | cg-coo --bench W,K) as a child of
    this rank's process: the row-partitioned path is implemented once, in C++ (HIPContext.cpp,
    comm*.cpp over RCCL); every rank of the launcher starts the same executable with its own
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, and rank 0's stdout carries the result.
    W untimed iterations, then K timed ones bracketed by a barrier + device synchronisation on
    both sides, the slowest rank's time (CGContextExt::run_fixed).

    If the job fails on ANY rank in its default form (the iteration replayed as a hipGraph, the scalar
    all-reduces over the peer board, halo windows through shared memory), ALL ranks -- they agree on it,
    agree_codes -- repeat it once in the most conservative form: eager enqueue, every collective an RCCL
    call (ABFT_CG_GRAPH=0, ABFT_COMM_ALLREDUCE=rccl, ABFT_COMM_EXCHANGE=rccl).  The failed attempt goes
    into the output line as `first_attempt` (every rank's status, rank 0's stderr tail) and the line says
    which form was measured: an abort in the fast path is then a finding in the record, never just a
    changed label."""
    exe = os.path.join(ROOT, "abft_sparse_cg_amd", "host", "cg-" + fmt)
    if not os.path.exists(exe):
        raise SystemExit("%s not built (make -C abft_sparse_cg_amd/host)" % exe)
    env = dict(os.environ)
    if profile:
        env["ABFT_BENCH_PROFILE"] = "1"
    cmd = [exe, "-t", "hip", "-m", mode, "-s", spec, "--bench", "%d,%d,%d" % (args.warmup, args.steps, BLOCKS), "-q"]
    rank = os.environ.get("RANK", "0")
    first_cmd = cmd
    if os.environ.get("ABFT_BENCH_INJECT_FAILURE") == rank:
        # test aid (tests/test_gpu_cli.py): this rank's first attempt dies at once, as a rank with a bad device would
        first_cmd = cmd + ["--no-such-option"]
    p = run_cpp(first_cmd, env)
    codes = agree_codes(p.returncode)
    replay = env.get("ABFT_CG_GRAPH", "1") != "0"
    first_attempt = None
    if any(codes) and replay:
        first_attempt = {"graph_replay": True, "returncodes_by_rank": codes, "rank0_stderr_tail": p.stderr[-1500:],
                         "second_attempt": "ABFT_CG_GRAPH=0 ABFT_COMM_ALLREDUCE=rccl ABFT_COMM_EXCHANGE=rccl",
                         "note": "status 70 = the first hipGraph replay did not finish within 180 s "
                                 "(HIPContext.cpp run_fixed); ranks that lose a peer end with other codes"}
        sys.stderr.write("rank %s: %s failed (statuses by rank: %s); all ranks once more, eagerly and over RCCL only\n%s\n"
                         % (rank, " ".join(cmd), codes, p.stderr[-1500:]))
        env.update(ABFT_CG_GRAPH="0", ABFT_COMM_ALLREDUCE="rccl", ABFT_COMM_EXCHANGE="rccl")
        replay = False
        p = run_cpp(cmd, env)
        codes = agree_codes(p.returncode)
    if any(codes):
        sys.stderr.write(p.stdout[-2000:] + p.stderr[-4000:])
        raise SystemExit("rank %s: %s exited with status %d (statuses by rank: %s)" % (rank, " ".join(cmd), p.returncode, codes))
    out = parse_cpp(p)
    out["graph_replay"] = replay
    if first_attempt:
        out["first_attempt"] = first_attempt
    return out


def one_rank_rr(args, spec, mode, fmt):
    """The same W + K iterations by ONE process on rank 0's GPU: the rr a multi-rank job must reproduce
    (to 1e-10: the shard sums group the additions differently) if its ranks solved the same system."""
    exe = os.path.join(ROOT, "abft_sparse_cg_amd", "host", "cg-" + fmt)
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ABFT_COMM_FORCE", "ABFT_HIP_GPUS", "ABFT_BENCH_PROFILE",
                        "TORCHELASTIC_USE_AGENT_STORE")}
    env.setdefault("ABFT_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0"))
    p = run_cpp([exe, "-t", "hip", "-m", mode, "-s", spec, "--bench", "%d,%d,%d" % (args.warmup, args.steps, BLOCKS), "-q"], env)
    if p.returncode != 0:
        return {"error": "one-rank run exited with status %d: %s" % (p.returncode, p.stderr[-600:])}
    return parse_cpp(p)


def block_stats(job, steps):
    """median / slowest / fastest block of a cg-* --bench W,K,B job -> keys for the bench line"""
    bs = job.get("block_seconds") or [job["seconds"]]
    return {"it_per_s": round(steps / job["seconds"], 2), "ms_per_step": round(job["seconds"] / steps * 1e3, 4),
            "blocks": len(bs), "it_per_s_min": round(steps / max(bs), 2), "it_per_s_max": round(steps / min(bs), 2),
            "value_is": "median of %d blocks of %d timed steps (every block: r = b, the warm-up steps, the timed steps)"
                        % (len(bs), steps),
            "iterations_per_block": job.get("iterations_per_block"), "rr_after_last_step": job["rr"]}


EXTRA_SPEC_MULTI = os.environ.get("ABFT_BENCH_EXTRA_SPEC", "random:4194304,24,1")  # (override: tests)


def multi(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, args.gpus))
    head = cpp_job(args, args.spec, args.mode, args.fmt, not args.no_profile)
    extra = None
    if args.extras:
        # the configuration the multi-GPU target is quoted on (BASELINE.json configs[3]): same path
        extra = cpp_job(args, EXTRA_SPEC_MULTI, "secded", "csr", not args.no_profile)
    if rank != 0:
        return None

    def leg(job, spec, fmt, mode):
        d = {"N": job["N"], "nnz": job["nnz"], "graph_replay": job["graph_replay"]}
        d.update(block_stats(job, args.steps))
        d["transport_by_rank"] = job.get("transport", [])
        if "first_attempt" in job:
            d["first_attempt"] = job["first_attempt"]
        if world > 1 and os.environ.get("ABFT_BENCH_RR_CHECK", "1") != "0":
            # self-validation: the same iterations by one process on rank 0's GPU
            one = one_rank_rr(args, spec, mode, fmt)
            if "rr" in one:
                rel = abs(job["rr"] - one["rr"]) / abs(one["rr"]) if one["rr"] else float("inf")
                d["rr_check"] = {"one_rank_rr": one["rr"], "rel_diff": rel, "tolerance": 1e-10, "ok": bool(rel <= 1e-10),
                                 "one_rank_it_per_s": round(args.steps / one["seconds"], 2)}
            else:
                d["rr_check"] = {"ok": False, "error": one.get("error", "no bench line")}
        sp = job.get("spmv")
        if sp and sp["spmvs"]:
            us = sp["total_us"] / sp["spmvs"]  # per SpMV of rank 0's shard (all its launches)
            byts = spmv_bytes(fmt, sp["rows"], sp["nnz"])
            d["spmv_rank0"] = {"avg_spmv_us": round(us, 2), "launches_per_spmv": sp["brackets"] / sp["spmvs"],
                               "rows": sp["rows"], "nnz": sp["nnz"], "algorithmic_bytes_per_spmv": byts,
                               "achieved": round(byts / us / 1e3, 1), "unit": "GB/s",
                               "frac": round(byts / us / 1e3 / HBM_PEAK_GBPS, 4)}
        return d
    return head, leg(head, args.spec, args.fmt, args.mode), (leg(extra, EXTRA_SPEC_MULTI, "csr", "secded") if extra else None)


def main():
    args = parse()
    base = {"metric": "CG iterations/sec + SpMV effective HBM GB/s (% of roofline), 1/2/4/8 MI355X",
            "unit": "CG iterations/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic"}
    if args.gpus <= 1 and int(os.environ.get("WORLD_SIZE", "1")) <= 1 and os.environ.get("ABFT_BENCH_SHARDED") != "1":
        dt, n, nnz, roof, kernels, cpu, probe, rr, extras, block_dt, spec_stats = single(args)
        out = dict(base)
        out.update({"value": round(args.steps / dt, 2), "ms_per_step": round(dt / args.steps * 1e3, 4),
                    "blocks": len(block_dt), "value_min": round(args.steps / max(block_dt), 2),
                    "value_max": round(args.steps / min(block_dt), 2),
                    "value_is": "median of %d blocks of %d timed steps (every block: r = b, the warm-up steps, the timed steps)"
                                % (len(block_dt), args.steps),
                    "config": {"workload": "cg-csr -t hip -m %s, synthetic %s" % (args.mode, args.spec)
                               if args.fmt == "csr" else "cg-coo -t hip -m %s, synthetic %s" % (args.mode, args.spec),
                               "N": n, "nnz": nnz, "format": args.fmt, "mode": args.mode, "parallelism": "1 GPU",
                               "rr_after_last_step": rr,
                               # every block restarts the solve: rr is the residual after warmup + steps iterations
                               "iterations_per_block": args.warmup + args.steps,
                               "speculated_iterations": spec_stats},
                    "roofline": roof, "cpu_baseline": cpu, "kernels": kernels})
        if probe:
            out["stream_probe"] = probe
        if extras:
            out["extra_legs"] = extras
        print(json.dumps(out))
        return
    res = multi(args)
    try:  # (the gloo group of agree_codes, if one was made)
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
    if res is not None:
        head, hl, xl = res
        out = dict(base)
        sp = hl.get("spmv_rank0")
        roof = None
        if sp:
            roof = {"bound": "hbm", "kernel": "spmv_%s_kernel<%s> (rank 0 shard, bracketed after the timed region)"
                    % (args.fmt, args.mode), "achieved": sp["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": sp["frac"], "traffic": None, "avg_launch_us": sp["avg_spmv_us"],
                    "algorithmic_bytes_per_launch": sp["algorithmic_bytes_per_spmv"]}
        out.update({"value": hl["it_per_s"], "ms_per_step": hl["ms_per_step"], "blocks": hl["blocks"],
                    "value_min": hl["it_per_s_min"], "value_max": hl["it_per_s_max"], "value_is": hl["value_is"],
                    "config": {"workload": "cg-%s -t hip -m %s, synthetic %s" % (args.fmt, args.mode, args.spec),
                               "N": hl["N"], "nnz": hl["nnz"], "format": args.fmt, "mode": args.mode,
                               "parallelism": "%d ranks (one process per GPU, C++ host over RCCL): output blocks cut by "
                                              "non-zeros, exchange of the search vector (all-gather: RCCL; halo windows: "
                                              "shared host memory) + 2 all-reduces per iteration (peer board: a copy per "
                                              "rank in IPC-mapped device memory, else in shared host memory, else RCCL -- "
                                              "whichever passes its start-up test; what each rank used: transport_by_rank), "
                                              "scalars device-resident, iteration %s" % (
                                                  args.gpus, "replayed as a hipGraph" if hl["graph_replay"] else
                                                  ("enqueued eagerly, every collective on the collective layer (the default form failed: first_attempt)"
                                                   if "first_attempt" in hl else "enqueued eagerly (ABFT_CG_GRAPH=0)")),
                               "rr_after_last_step": hl["rr_after_last_step"],
                               "iterations_per_block": hl["iterations_per_block"]},
                    "roofline": roof, "cpu_baseline": None,
                    # the record of an unattended run: what carried it, and that its ranks solved the same system
                    "transport_by_rank": hl["transport_by_rank"], "rr_check": hl.get("rr_check")})
        if "first_attempt" in hl:
            out["first_attempt"] = hl["first_attempt"]
        if xl:
            xl["workload"] = "cg-csr -t hip -m secded, synthetic %s (BASELINE.json configs[3])" % EXTRA_SPEC_MULTI
            out["extra_legs"] = {"config4": xl}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
