#!/bin/bash
# A/B of the slice layout against the sweep layout on config 4 (and its 1/8-shard-sized cousin):
# one bench.py process per variant, SpMV time from the HIP-event brackets.
# usage: tools/slice_ab.sh [mode] ; env VARIANTS="name:ENV=V,ENV=V ..." adds library/env variants
O=gpurun_out/r3; mkdir -p $O
MODE=${1:-secded}
SPEC=${SPEC:-random:4194304,24,1}
one() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ] && [ -n "$1" ]; do envs+=("$1"); shift; done
  env "${envs[@]}" python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 30 --warmup 5 --spec $SPEC --mode $MODE > $O/sl_$name.json 2> $O/sl_$name.err
  python3 -c "
import json,sys
try:
    d=json.load(open('$O/sl_$name.json')); print('%-34s spmv %8.1f us   %7.1f it/s' % ('$name', d['kernels']['spmv']['avg_us'], d['value']))
except Exception as e:
    print('%-34s FAILED %r' % ('$name', e)); print(open('$O/sl_$name.err').read()[-600:])"
}
for v in "$@"; do :; done
if [ -z "$SLICE_ONLY" ]; then one sweep ABFT_HIP_LAYOUT=sweep --; fi
for cfg in ${CFGS:-512:131072 1024:131072 256:131072 512:65536 512:262144}; do
  r=${cfg%%:*}; w=${cfg##*:}
  one slice_r${r}_w${w} ABFT_HIP_LAYOUT=slice ABFT_HIP_SLICE_ROWS=$r ABFT_HIP_PANEL_WIDTH=$w --
done
