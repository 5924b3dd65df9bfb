#!/bin/bash
# layouts of the 1/8 shard of config 4 (what a rank of the 8-GPU job multiplies), on one GPU
O=gpurun_out/r2; mkdir -p $O
one() { name=$1; shift; env "$@" python3 tools/shard_budget.py --ranks 8 2>/dev/null | grep "G=8 rank 0" | sed "s/^measured G=8 rank 0/$name/"; }
one stream ABFT_HIP_LAYOUT=stream
one panels ABFT_HIP_LAYOUT=panels
for r in 1 2 4; do for w in 131072 262144 524288; do
  one sweep_r${r}_w${w} ABFT_HIP_LAYOUT=sweep ABFT_HIP_SWEEP_RPT=$r ABFT_HIP_PANEL_WIDTH=$w
done; done
