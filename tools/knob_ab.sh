#!/bin/bash
# bench.py's bracketed SpMV average under sets of environment knobs: knob_ab.sh "<bench args>" "A=1 B=2" "A=2" ...
O=gpurun_out/r2; mkdir -p $O
ARGS=$1; shift
for kv in "$@"; do
  env $kv python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 40 --warmup 5 $ARGS 2>/dev/null |
    python3 -c "import json,sys; d=json.load(sys.stdin); print('%-60s' % '$kv', d['kernels']['spmv'], 'it/s %.1f' % d['value'])"
done
