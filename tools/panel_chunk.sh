#!/bin/bash
A="--steps 20 --mode none --spec random:4194304,24,1"
for ch in 0 1 2 4; do echo "chunk $ch"; ABFT_HIP_PANEL_CHUNK=$ch tools/ab_variants.sh "$A" base; done
echo "chunk 1 width 2^17"; ABFT_HIP_PANEL_WIDTH=131072 ABFT_HIP_PANEL_CHUNK=1 tools/ab_variants.sh "$A" base
echo "chunk 1 width 2^19"; ABFT_HIP_PANEL_WIDTH=524288 ABFT_HIP_PANEL_CHUNK=1 tools/ab_variants.sh "$A" base
ABFT_HIP_PANEL_CHUNK=1 tools/pmc_tcc.sh chunk1 "--mode none --spec random:4194304,24,1"
ABFT_HIP_LAYOUT=panels ABFT_HIP_PANEL_WIDTH=16 ABFT_HIP_PANEL_CHUNK=3 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
