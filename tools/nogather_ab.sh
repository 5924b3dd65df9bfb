#!/bin/bash
O=gpurun_out/r2; mkdir -p $O
one() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 30 --warmup 5 --spec random:4194304,24,1 "$@" > $O/ng_$name.json 2> $O/ng_$name.err
  python3 -c "
import json,sys
d=json.load(open('$O/ng_$name.json')); print('%-28s spmv %8.1f us' % ('$name', d['kernels']['spmv']['avg_us']))"
}
NG=ABFT_HIP_LIB=$PWD/variants/lib_NOGATHER.so
one stream_none ABFT_HIP_LAYOUT=stream -- --mode none
one stream_nogather_none $NG ABFT_HIP_LAYOUT=stream -- --mode none
one stream_nogather_secded $NG ABFT_HIP_LAYOUT=stream -- --mode secded
for r in 8 16; do
one sweep_r${r}_none ABFT_HIP_SWEEP_RPT=$r -- --mode none
one sweep_r${r}_nogather_none $NG ABFT_HIP_SWEEP_RPT=$r -- --mode none
one sweep_r${r}_nogather_none_lag0 $NG ABFT_HIP_SWEEP_RPT=$r ABFT_HIP_SWEEP_LAG=0 -- --mode none
one sweep_r${r}_nogather_none_w18 $NG ABFT_HIP_SWEEP_RPT=$r ABFT_HIP_PANEL_WIDTH=262144 -- --mode none
one sweep_r${r}_nogather_none_w20 $NG ABFT_HIP_SWEEP_RPT=$r ABFT_HIP_PANEL_WIDTH=1048576 -- --mode none
done
