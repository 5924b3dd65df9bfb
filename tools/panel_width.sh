#!/bin/bash
for w in 262144 131072 65536 32768 16384; do
  echo "width $w"
  ABFT_HIP_PANEL_WIDTH=$w tools/ab_variants.sh "--steps 20 --mode none --spec random:4194304,24,1" base
done
