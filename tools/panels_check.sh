#!/bin/bash
O=gpurun_out/panels_check.log
: > $O
ABFT_HIP_LAYOUT=panels ABFT_HIP_PANEL_WIDTH=300 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q >> $O 2>&1; echo "rc=$?" >> $O
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -x -q >> $O 2>&1; echo "rc=$?" >> $O
tail -12 $O
tools/ab_variants.sh "--steps 30 --mode secded --spec random:4194304,24,1" base
ABFT_HIP_LAYOUT=stream tools/ab_variants.sh "--steps 30 --mode secded --spec random:4194304,24,1" base
tools/ab_variants.sh "--steps 30 --mode none --spec random:4194304,24,1" base
