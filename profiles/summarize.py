#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the per-kernel summaries kept under profiles/.

    python profiles/summarize.py trace  <dir> <out.md>      # --kernel-trace --stats run
    python profiles/summarize.py pmc    <dir> <out.json>    # one --pmc run (any counters)

trace: per kernel name, launches / total / average / min / max duration (ns from
the kernel-trace CSV's Start_Timestamp / End_Timestamp).
pmc:   per kernel name and counter, the average value per launch.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace(" [clone .kd]", "").replace(".kd", "")
    return name.strip()


def find(d, pattern):
    return sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))


def trace(d, out):
    rows = defaultdict(list)
    for f in find(d, "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            rows[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in rows.values()) or 1
    with open(out, "w") as o:
        o.write("| kernel | launches | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            o.write("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |\n" % (
                k, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))
    print(open(out).read())


def pmc(d, out):
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: {c: {"avg_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()}
           for k, cs in acc.items()}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, cs in res.items():
        for c, v in cs.items():
            print("%-60s %-18s %14.1f  (%d launches)" % (k[:60], c, v["avg_per_launch"], v["launches"]))


def pmc_phases(d, out, pattern, k):
    """per-launch counters of a kernel that one SpMV launches `k` times (the panel layouts: a launch per
    chunk of panels): dispatches matching `pattern`, in dispatch order, dealt to phase 0 .. k-1"""
    k = int(k)
    rows = []
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if pattern in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
    per = defaultdict(lambda: defaultdict(float))  # dispatch -> counter -> value (summed over the CSV's rows)
    for disp, c, v in rows:
        per[disp][c] += v
    acc = defaultdict(lambda: defaultdict(list))
    for n, disp in enumerate(sorted(per)):
        for c, v in per[disp].items():
            acc[n % k][c].append(v)
    res = {"phase %d" % ph: {c: {"avg_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()}
           for ph, cs in acc.items()}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for ph in sorted(res):
        for c, v in res[ph].items():
            print("%-10s %-18s %14.1f  (%d launches)" % (ph, c, v["avg_per_launch"], v["launches"]))


if __name__ == "__main__":
    if sys.argv[1] == "pmc_phases":
        pmc_phases(*sys.argv[2:6])
    else:
        {"trace": trace, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
