#!/bin/bash
O=gpurun_out/panels_check.log
: > $O
for w in 16 300; do for ch in 0 3; do
  ABFT_HIP_LAYOUT=panels ABFT_HIP_PANEL_WIDTH=$w ABFT_HIP_PANEL_CHUNK=$ch timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q >> $O 2>&1; echo "width $w chunk $ch rc=$?" >> $O
done; done
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -x -q >> $O 2>&1; echo "fullsize rc=$?" >> $O
grep -E "rc=|passed|failed" $O
tools/ab_variants.sh "--steps 30 --mode secded --spec random:4194304,24,1" base
tools/ab_variants.sh "--steps 30 --mode none --spec random:4194304,24,1" base
tools/ab_variants.sh "--steps 50 --fmt coo --mode sec7 --spec powerlaw:2097152,2" base
