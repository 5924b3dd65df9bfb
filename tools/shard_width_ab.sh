#!/bin/bash
# 1/8 row shard of config 4 (what a rank of the 8-GPU job multiplies): panel widths of the sweep layout
O=gpurun_out/r2; mkdir -p $O
one() { name=$1; shift; env "$@" python3 tools/shard_budget.py --ranks ${RANKS:-8} 2>/dev/null | grep "G=${RANKS:-8} rank 0" | sed "s/^measured G=${RANKS:-8} rank 0/$name/"; }
for w in ${WIDTHS:-262144 290000 305000 320000 335000}; do
  one sweep_w${w} ABFT_HIP_LAYOUT=sweep ABFT_HIP_PANEL_WIDTH=$w
done
