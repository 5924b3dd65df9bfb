#!/bin/bash
# usage: tools/pmc_one.sh <tag> "<bench args>" : FETCH_SIZE + TCC hit/miss of every kernel, summarized
export TMPDIR=/tmp
O=gpurun_out/pmc_$1
rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE -f csv -d $O/f -- python3 bench.py --cpu-iters 0 --steps 4 --warmup 1 --no-profile $2 > /dev/null 2> $O/f.err
python3 profiles/summarize.py pmc $O/f $O/fetch.json | grep -i spmv
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -f csv -d $O/t -- python3 bench.py --cpu-iters 0 --steps 4 --warmup 1 --no-profile $2 > /dev/null 2> $O/t.err
python3 profiles/summarize.py pmc $O/t $O/tcc.json | grep -i spmv
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -f csv -d $O/s -- python3 bench.py --cpu-iters 0 --steps 4 --warmup 1 --no-profile $2 > /dev/null 2> $O/s.err
python3 profiles/summarize.py pmc $O/s $O/sq.json | grep -i spmv
rm -rf $O/f $O/t $O/s
