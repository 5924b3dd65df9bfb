#!/bin/bash
# memory latency / concurrency counters of the SpMV for two library builds (base and variants/lib_$1.so)
export TMPDIR=/tmp
O=gpurun_out/pmclat
rm -rf $O; mkdir -p $O
for v in base $1; do
  if [ "$v" = base ]; then unset ABFT_HIP_LIB; else export ABFT_HIP_LIB=$PWD/variants/lib_$v.so; fi
  i=0
  for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $set -f csv -d $O/$v$i -- python3 bench.py --cpu-iters 0 --steps 4 --warmup 1 --no-profile > /dev/null 2> $O/$v$i.err
    echo "== $v: $set"
    python3 profiles/summarize.py pmc $O/$v$i $O/$v$i.json | grep -i "spmv" | cut -c 1-120
    rm -rf $O/$v$i
  done
done
