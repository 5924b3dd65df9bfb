#!/bin/bash
for spec in "random:262144,24,1" "random:4194304,24,1" "laplace5:3162,3162"; do
  echo "== $spec (auto layout)"
  tools/ab_variants.sh "--steps 30 --mode none --spec $spec" base gnt gsc1
done
echo "== random 4M stream layout"
ABFT_HIP_LAYOUT=stream tools/ab_variants.sh "--steps 20 --mode none --spec random:4194304,24,1" base gnt gsc1
