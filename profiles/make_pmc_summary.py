#!/usr/bin/env python3
"""profiles/r04/pmc_*.json (ABFT_PROFILE_ROUND picks another round) (separate rocprofv3 --pmc passes, condensed by summarize.py)
-> profiles/pmc_summary.json: HBM bytes per SpMV, which bench.py reports as
roofline.traffic.

Per the MI355X guide's HBM / rocprofv3 section: FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read, so it
is doubled -- and the factor is checked here on the vector kernels of the same run,
whose read bytes are known exactly (calc_r reads 16 N bytes, calc_px 24 N).  A panel
layout SpMV is several launches: bytes per SpMV = per-launch average x launches per SpMV
(the fold kernel runs once per SpMV and gives the count).

    python profiles/make_pmc_summary.py            (from the repo root)
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
R = os.path.join(HERE, os.environ.get("ABFT_PROFILE_ROUND", "r04"))
CASES = {  # tag -> (workload key used by bench.py, bench line of the traced run)
    "laplace_none": ("laplace5:3162,3162/csr/none", "bench_under_trace_laplace_none.json"),
    "laplace_sed": ("laplace5:3162,3162/csr/sed", "bench_under_trace_laplace_sed.json"),
    "laplace_secded": ("laplace5:3162,3162/csr/secded", "bench_under_trace_laplace_secded.json"),
    "random_secded": ("random:4194304,24,1/csr/secded", "bench_under_trace_random_secded.json"),
    "powerlaw_coo_sec7": ("powerlaw:2097152,2/coo/sec7", "bench_under_trace_powerlaw_coo_sec7.json"),
}


def load(name):
    return json.load(open(os.path.join(R, name)))


def pick(d, word):
    ks = [k for k in d if word in k]
    return ks[0] if ks else None


def main():
    out = {}
    for tag, (workload, benchfile) in CASES.items():
        f, w, t = (load("pmc_%s_%s.json" % (c, tag)) for c in ("fetch_size", "write_size", "tcc_hit_sum"))
        n = json.load(open(os.path.join(R, benchfile)))["config"]["N"]
        sps = [k for k in f if "spmv" in k]  # the panel layout runs two instantiations (with / without the fused dot)
        fold = pick(f, "fuse_finalize") or pick(f, "fold_partials")
        n_spmv = max(f[fold]["FETCH_SIZE"]["launches"], 1) if fold else f[sps[0]]["FETCH_SIZE"]["launches"]

        def per_spmv_total(d, counter):
            return sum(d[k][counter]["avg_per_launch"] * d[k][counter]["launches"] for k in sps) / n_spmv
        per_spmv = sum(f[k]["FETCH_SIZE"]["launches"] for k in sps) / n_spmv
        fetch = per_spmv_total(f, "FETCH_SIZE")
        write = per_spmv_total(w, "WRITE_SIZE")
        calib = {}
        for kern, nbytes in (("calc_r_kernel", 16 * n), ("calc_px_kernel", 24 * n)):
            k = pick(f, kern)
            if k:
                calib[kern] = round(nbytes / (f[k]["FETCH_SIZE"]["avg_per_launch"] * 1024.0), 3)
        hit = per_spmv_total(t, "TCC_HIT_sum")
        miss = per_spmv_total(t, "TCC_MISS_sum")
        out[workload] = {
            "FETCH_SIZE_KiB_raw_per_spmv": fetch,
            "WRITE_SIZE_KiB_raw_per_spmv": write,
            "kernel_launches_per_spmv": per_spmv,
            "TCC_hit_rate": hit / (hit + miss),
            "fetch_factor_measured_on_vector_kernels": calib,
            "spmv_hbm_bytes_per_launch": int(round((2.0 * fetch + write) * 1024.0)),
            "note": "per SpMV; FETCH_SIZE x2 (gfx950 correction for wide coalesced reads; the factor measured on "
                    "this run's vector kernels is listed) + WRITE_SIZE; for the panel layout (several launches per "
                    "SpMV) x2 is an upper bound: 8-byte gathers are not wide streams",
        }
    json.dump(out, open(os.path.join(HERE, "pmc_summary.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
