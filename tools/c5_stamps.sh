#!/bin/bash
# config 5 (COO power-law, sec7): where a workgroup of spmv_coo_panels_kernel spends its time.
#   base      the shipped library (SpMV time by HIP events inside bench.py)
#   STAMPS    -DABFT_DBG_STAMPS build (variants/lib_STAMPS.so): wave 0's clock per phase, printed at matrix destruction
#   NOGATHER  -DABFT_DBG_NOGATHER build: the same kernel without its x gathers (wrong results, timing only)
# usage: tools/c5_stamps.sh [variant...]     (default: base STAMPS NOGATHER)
O=gpurun_out/c5_stamps
mkdir -p $O
SPEC=${SPEC:-powerlaw:2097152,2}
MODE=${MODE:-sec7}
for v in ${@:-base STAMPS NOGATHER}; do
  if [ "$v" = base ]; then unset ABFT_HIP_LIB; else export ABFT_HIP_LIB=$PWD/variants/lib_$v.so; fi
  ABFT_HIP_PANEL_DEBUG=1 python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 40 --warmup 4 --fmt coo --mode $MODE \
      --spec $SPEC > $O/$v.json 2> $O/$v.err
  echo "== $v: $(python3 -c "import json,sys; d=json.load(open('$O/$v.json')); print('spmv', d['kernels']['spmv']['avg_us'], 'us,', d['value'], 'it/s')")"
  grep "panel phases" $O/$v.err
done
