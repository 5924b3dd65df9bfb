#!/bin/bash
# sweep layout, config 4: panel widths that fill the last tile of a segment (not only powers of two)
O=gpurun_out/r2; mkdir -p $O
SPEC=${SPEC:---mode secded --spec random:4194304,24,1}
for lag in ${LAGS:-2}; do for w in ${WIDTHS:-131072 150000 156000 160000 78000 236000 262144}; do
  ABFT_HIP_LAYOUT=sweep ABFT_HIP_PANEL_WIDTH=$w ABFT_HIP_SWEEP_LAG=$lag python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 40 --warmup 5 $SPEC 2>$O/width_$w.err |
    python3 -c "import json,sys; d=json.load(sys.stdin); print('lag $lag width $w', d['kernels']['spmv'], 'it/s %.1f' % d['value'])"
done; done
