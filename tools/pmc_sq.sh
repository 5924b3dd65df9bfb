#!/bin/bash
# usage: tools/pmc_sq.sh <tag> "<bench args>" : where the SpMV's wave-cycles go (separate --pmc passes)
# (the four TA_*_sum counters in ONE pass are more than the hardware collects at once: rocprofv3 aborts under
# abft_hip_init with "rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of
# the hardware to collect" -- gpurun_out/pmcsq_slice/p3.err, round 3; two passes of two since round 4.  A set
# that is rejected all the same is reported and skipped; the script is never looped.)
export TMPDIR=/tmp
O=gpurun_out/pmcsq_$1
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "MemUnitStalled MeanOccupancyPerCU" ; do
  i=$((i+1))
  rocprofv3 --pmc $set -f csv -d $O/p$i -- python3 bench.py --cpu-iters 0 --steps 4 --warmup 1 --no-profile --no-probe --no-extras $2 > /dev/null 2> $O/p$i.err
  if grep -q "error code 38" $O/p$i.err; then echo "pass $i ($set): rejected by rocprofv3 (error 38), skipped"; rm -rf $O/p$i; continue; fi
  python3 profiles/summarize.py pmc $O/p$i $O/p$i.json | grep -i "spmv" || echo "pass $i ($set): no spmv rows"
  rm -rf $O/p$i
done
