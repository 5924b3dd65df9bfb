#!/bin/bash
# COO constraints mode against `none` (bench.py's bracketed SpMV average), config 5's matrix and the Laplacian, for one or more builds
for lib in "${@:-}"; do
  for spec in powerlaw:2097152,2 laplace5:3162,3162; do for m in none constraints; do
    ABFT_HIP_LIB=$lib python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 30 --warmup 4 --fmt coo --mode $m --spec $spec 2>/dev/null |
      python3 -c "import json,sys; d=json.load(sys.stdin); print('%-26s %-22s %-12s' % ('${lib:-in-tree}', '$spec', '$m'), d['kernels']['spmv'])"
  done; done
done
