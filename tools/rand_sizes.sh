#!/bin/bash
for n in 65536 262144 524288 1048576 2097152 4194304; do
  echo "random N=$n"
  ABFT_HIP_LAYOUT=stream tools/ab_variants.sh "--steps 30 --mode none --spec random:$n,24,1" base
done
