#!/bin/bash
# config 5 (COO power-law, panel layout): panel widths that are not powers of two
O=gpurun_out/r2; mkdir -p $O
for w in ${WIDTHS:-100000 150000 200000 262144 300000 350000}; do for ch in ${CHUNKS:-4}; do
  ABFT_HIP_PANEL_WIDTH=$w ABFT_HIP_PANEL_CHUNK=$ch python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 40 --warmup 5 --fmt coo --mode sec7 --spec powerlaw:2097152,2 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('width $w chunk $ch', d['kernels']['spmv'])"
done; done
