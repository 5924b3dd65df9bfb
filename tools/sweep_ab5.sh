#!/bin/bash
O=gpurun_out/r2; mkdir -p $O
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 40 --warmup 5 "$@" > $O/sw_$name.json 2> $O/sw_$name.err
  python3 - "$name" $O/sw_$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); k=d["kernels"]["spmv"]
    print("%-28s spmv %8.1f us  %6.1f GB/s  it/s %8.1f" % (sys.argv[1], k["avg_us"], k["GBps"], d["value"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
C4="--mode secded --spec random:4194304,24,1"
C5="--fmt coo --mode sec7 --spec powerlaw:2097152,2"
run c4_panels ABFT_HIP_LAYOUT=panels -- $C4
for lag in 2 3; do for rpt in 8 16; do
  run c4_w262144_l${lag}_r${rpt} ABFT_HIP_LAYOUT=sweep ABFT_HIP_PANEL_WIDTH=262144 ABFT_HIP_SWEEP_LAG=$lag ABFT_HIP_SWEEP_RPT=$rpt -- $C4
done; done
run c4_w131072_l2_r16 ABFT_HIP_LAYOUT=sweep ABFT_HIP_PANEL_WIDTH=131072 ABFT_HIP_SWEEP_LAG=2 ABFT_HIP_SWEEP_RPT=16 -- $C4
run c4_w131072_l3_r16 ABFT_HIP_LAYOUT=sweep ABFT_HIP_PANEL_WIDTH=131072 ABFT_HIP_SWEEP_LAG=3 ABFT_HIP_SWEEP_RPT=16 -- $C4
run c5_panels ABFT_HIP_LAYOUT=panels -- $C5
run c5_w262144_l0_r8 ABFT_HIP_LAYOUT=sweep ABFT_HIP_PANEL_WIDTH=262144 ABFT_HIP_SWEEP_LAG=0 ABFT_HIP_SWEEP_RPT=8 -- $C5
run c5_w262144_l4_r8 ABFT_HIP_LAYOUT=sweep ABFT_HIP_PANEL_WIDTH=262144 ABFT_HIP_SWEEP_LAG=4 ABFT_HIP_SWEEP_RPT=8 -- $C5
