#!/bin/bash
# kernel-trace SpMV durations of configs 2 (secded), 4 and 5 for base and the given variants
export TMPDIR=/tmp
O=gpurun_out/trace_cfg
rm -rf $O; mkdir -p $O
for v in base "$@"; do
  if [ "$v" = base ]; then unset ABFT_HIP_LIB; else export ABFT_HIP_LIB=$PWD/variants/lib_$v.so; fi
  k=0
  for args in "--mode secded" "--steps 30 --mode secded --spec random:4194304,24,1" "--steps 50 --fmt coo --mode sec7 --spec powerlaw:2097152,2" "--steps 50 --fmt coo --mode none"; do
    k=$((k+1))
    rocprofv3 --kernel-trace --stats -f csv -d $O/$v$k -- python3 bench.py --cpu-iters 0 --no-profile $args > $O/$v$k.json 2> $O/$v$k.err
    echo "== $v: $args"; python3 profiles/summarize.py trace $O/$v$k $O/$v$k.md | grep spmv
    rm -rf $O/$v$k
  done
done
