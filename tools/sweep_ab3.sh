#!/bin/bash
# where the sweep kernel's time goes: timing-only variant builds and modes (config 4)
O=gpurun_out/r2; mkdir -p $O
export ABFT_HIP_LAYOUT=sweep ABFT_HIP_PANEL_WIDTH=262144 ABFT_HIP_SWEEP_LAG=2 ABFT_HIP_SWEEP_RPT=16
one() {
  name=$1; shift
  python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 30 --warmup 5 --spec random:4194304,24,1 "$@" > $O/v_$name.json 2> $O/v_$name.err
  python3 - "$name" $O/v_$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); k=d["kernels"]["spmv"]
    print("%-24s spmv %8.1f us" % (sys.argv[1], k["avg_us"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
one base_secded --mode secded
one base_none --mode none
one base_sed --mode sed
ABFT_HIP_LIB=$PWD/variants/lib_NOGATHER.so one nogather_secded --mode secded
ABFT_HIP_LIB=$PWD/variants/lib_NOGATHER.so one nogather_none --mode none
ABFT_HIP_LIB=$PWD/variants/lib_NOPHASE2.so one nophase2_secded --mode secded
ABFT_HIP_LIB=$PWD/variants/lib_NOPHASE2.so one nophase2_none --mode none
ABFT_HIP_SWEEP_RPT=8 one r8_secded --mode secded
ABFT_HIP_SWEEP_RPT=8 one r8_none --mode none
