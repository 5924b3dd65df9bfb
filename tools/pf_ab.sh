#!/bin/bash
O=gpurun_out/r2; mkdir -p $O
one() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 30 --warmup 5 --spec random:4194304,24,1 "$@" > $O/pf_$name.json 2> $O/pf_$name.err
  python3 -c "
import json,sys
d=json.load(open('$O/pf_$name.json')); print('%-28s spmv %8.1f us' % ('$name', d['kernels']['spmv']['avg_us']))"
}
PF1=ABFT_HIP_LIB=$PWD/variants/lib_PF1.so
PF3=ABFT_HIP_LIB=$PWD/variants/lib_PF3.so
one base_r16_secded ABFT_HIP_SWEEP_RPT=16 -- --mode secded
one base_r8_secded ABFT_HIP_SWEEP_RPT=8 -- --mode secded
one pf_r8_secded $PF1 ABFT_HIP_SWEEP_RPT=8 -- --mode secded
one pf_r8_none $PF1 ABFT_HIP_SWEEP_RPT=8 -- --mode none
one pf_r8_secded_w18 $PF1 ABFT_HIP_SWEEP_RPT=8 ABFT_HIP_PANEL_WIDTH=262144 -- --mode secded
one pf_r16_secded $PF3 ABFT_HIP_SWEEP_RPT=16 -- --mode secded
one pf_r8_secded_lag3 $PF1 ABFT_HIP_SWEEP_RPT=8 ABFT_HIP_SWEEP_LAG=3 -- --mode secded
