#!/bin/bash
# PMC passes (separate runs, as the guide prescribes) for one bench.py workload:
#   tools/pmc_one.sh NAME ENV... -- BENCH_ARGS...   -> gpurun_out/${ROUND:-r3}/pmc_NAME_*.json
export TMPDIR=/tmp
O="gpurun_out/${ROUND:-r3}"; mkdir -p "$O"
name=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
for e in "${envs[@]}"; do export "$e"; done
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1 | tr A-Z a-z)
  rm -rf "$O/raw_${name}_${n}"
  rocprofv3 --pmc $c -f csv -d "$O/raw_${name}_${n}" -- python3 bench.py --cpu-iters 0 --no-probe --no-extras --no-profile --steps 5 --warmup 1 "$@" > /dev/null 2> "$O/pmc_${name}_${n}.err"
  python3 profiles/summarize.py pmc "$O/raw_${name}_${n}" "$O/pmc_${name}_${n}.json" | grep -i -E "spmv|sweep|slice|fixup"
  rm -rf "$O/raw_${name}_${n}"
done
