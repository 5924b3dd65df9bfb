#!/bin/bash
# COO SpMV by ECC mode (bench.py's bracketed average) on config 5 and on the Laplacian, for one or more builds:
#   coo_modes.sh [lib.so ...]    (default: the in-tree build)
O=gpurun_out/r2; mkdir -p $O
for lib in "${@:-}"; do
  for spec in powerlaw:2097152,2 laplace5:3162,3162; do for m in none constraints sed sec7 sec8 secded; do
    ABFT_HIP_LIB=$lib python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 30 --warmup 4 --fmt coo --mode $m --spec $spec 2>/dev/null |
      python3 -c "import json,sys; d=json.load(sys.stdin); print('%-22s %-22s %-12s' % ('${lib:-in-tree}', '$spec', '$m'), d['kernels']['spmv'])"
  done; done
done
