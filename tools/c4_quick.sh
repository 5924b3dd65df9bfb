#!/bin/bash
# config 4 (and its 1/8 shard) SpMV, quick
for m in secded none; do
python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 40 --warmup 5 --spec random:4194304,24,1 --mode $m 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('config4 $m', d['kernels']['spmv'])"
done
python3 tools/shard_budget.py --ranks 8 2>/dev/null | grep "G=8 rank 0"
