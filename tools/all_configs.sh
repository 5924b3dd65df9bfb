#!/bin/bash
# one bench.py line per BASELINE.json single-GPU configuration (2, 3, 4 on one GPU, 5)
O=gpurun_out/all_configs.jsonl
: > $O
python bench.py --cpu-iters 80 >> $O 2>/dev/null
python bench.py --cpu-iters 80 --mode sed >> $O 2>/dev/null
python bench.py --cpu-iters 80 --mode secded >> $O 2>/dev/null
python bench.py --cpu-iters 24 --steps 100 --mode secded --spec random:4194304,24,1 >> $O 2>/dev/null
python bench.py --cpu-iters 0 --steps 100 --fmt coo --mode sec7 --spec powerlaw:2097152,2 >> $O 2>/dev/null
python bench.py --cpu-iters 0 --steps 100 --fmt coo --mode none >> $O 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/all_configs.jsonl"):
    d = json.loads(l)
    r, c = d["roofline"], d["cpu_baseline"]
    print("%-66s %8.1f it/s  SpMV %6.1f us %6.0f GB/s (%.0f %% of 8 TB/s)  CPU %s" % (
        d["config"]["workload"], d["value"], r["avg_launch_us"], r["achieved"], 100 * r["frac"],
        "%.1f it/s on %d cores, %.1f on 1" % (c["value"], c["cores"], c["one_core"]["value"]) if c else "-"))
PY
