#!/bin/bash
# the iteration's tail as one launch (default) against its three kernels (ABFT_CG_TAIL=0): cg-csr --bench, one process
H=abft_sparse_cg_amd/host
for spec in "laplace5:3162,3162 none" "random:4194304,24,1 secded" "laplace5:1000,1000 secded" "random:524288,24,1 secded"; do
  set -- $spec
  for t in 1 0 1 0; do
    echo -n "$1 $2 ABFT_CG_TAIL=$t: "
    ABFT_CG_TAIL=$t $H/cg-csr -t hip -m $2 -s $1 --bench 20,200,5 -q | grep "^bench:" | awk '{print $11, "it/s"}'
  done
done
