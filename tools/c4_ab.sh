#!/bin/bash
# config 4 (sweep kernel): SpMV time of the shipped library against variant builds, alternated
run() { env "$@" python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 30 --warmup 4 --mode ${MODE:-secded} --spec ${SPEC:-random:4194304,24,1} 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['kernels']['spmv']['avg_us'])"; }
for i in 1 2; do for v in base "$@"; do
  if [ "$v" = base ]; then echo -n "base: "; run X=1; else echo -n "$v: "; run ABFT_HIP_LIB=$PWD/variants/lib_$v.so; fi
done; done
