#!/bin/bash
# usage: tools/pmc_tcc.sh <tag> "<bench args>" : TCC hit/miss of the SpMV kernel
export TMPDIR=/tmp
O=gpurun_out/pmc_$1
rm -rf $O; mkdir -p $O
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $O/t -- python3 bench.py --cpu-iters 0 --steps 3 --warmup 1 --no-profile $2 > /dev/null 2> $O/t.err
echo "== $1"; python3 profiles/summarize.py pmc $O/t $O/tcc.json | grep -i "spmv"
rm -rf $O/t
