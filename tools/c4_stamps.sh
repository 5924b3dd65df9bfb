#!/bin/bash
# config 4 (random 2^22 x 24, sweep layout): where the workgroups of spmv_sweep_kernel spend their time, by mode and
# pacing lag -- wave 0's clock per phase and the per-workgroup spread (variants/lib_STAMPS.so, ABFT_HIP_SWEEP_DEBUG=1)
O=gpurun_out/c4_stamps
mkdir -p $O
SPEC=${SPEC:-random:4194304,24,1}
for mode in none secded; do for lag in ${LAGS:-2 3 0}; do
  for v in base STAMPS; do
    if [ "$v" = base ]; then unset ABFT_HIP_LIB; else export ABFT_HIP_LIB=$PWD/variants/lib_$v.so; fi
    ABFT_HIP_SWEEP_DEBUG=1 ABFT_HIP_SWEEP_LAG=$lag python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 30 --warmup 4 --mode $mode \
      --spec $SPEC > $O/$v.json 2> $O/$v.err
    echo "== $mode lag $lag $v: $(python3 -c "import json; d=json.load(open('$O/$v.json')); print('spmv', d['kernels']['spmv']['avg_us'], 'us')")"
    grep "sweep p\|sweep w\|per XCD" $O/$v.err
  done
done; done
