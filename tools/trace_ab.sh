#!/bin/bash
# kernel-trace durations (no counters) of bench.py for base and the given variants
# (variants/lib_<name>.so); BENCH_ARGS = extra bench.py arguments (workload)
export TMPDIR=/tmp
O=gpurun_out/trace_ab
rm -rf $O; mkdir -p $O
for v in base "$@"; do
  if [ "$v" = base ]; then unset ABFT_HIP_LIB; else export ABFT_HIP_LIB=$PWD/variants/lib_$v.so; fi
  rocprofv3 --kernel-trace --stats -f csv -d $O/$v -- python3 bench.py --cpu-iters 0 --steps 100 --no-profile --no-probe $BENCH_ARGS > $O/$v.json 2> $O/$v.err
  echo "== $v"; python3 profiles/summarize.py trace $O/$v $O/$v.md | head -8
  rm -rf $O/$v
done
