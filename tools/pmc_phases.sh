#!/bin/bash
# Per-launch counters of a panel-layout SpMV (one SpMV = K launches of the same kernel):
#   tools/pmc_phases.sh NAME KERNEL_SUBSTRING K -- BENCH_ARGS...   -> gpurun_out/${ROUND:-r3}/pmcph_NAME_*.json
export TMPDIR=/tmp
O="gpurun_out/${ROUND:-r3}"; mkdir -p "$O"
name=$1; pat=$2; k=$3; shift 3; [ "$1" = "--" ] && shift
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=$(echo $c | cut -d' ' -f1 | tr A-Z a-z)
  rm -rf "$O/raw_${name}_${n}"
  rocprofv3 --pmc $c -f csv -d "$O/raw_${name}_${n}" -- python3 bench.py --cpu-iters 0 --no-probe --no-extras --no-profile --steps 5 --warmup 1 "$@" > /dev/null 2> "$O/pmcph_${name}_${n}.err"
  echo "== $c"; python3 profiles/summarize.py pmc_phases "$O/raw_${name}_${n}" "$O/pmcph_${name}_${n}.json" "$pat" "$k"
  rm -rf "$O/raw_${name}_${n}"
done
