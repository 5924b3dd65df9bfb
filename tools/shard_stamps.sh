#!/bin/bash
# the 1/8 shard of config 4 (rank 0 of 8) on one GPU: SpMV time of the shipped library and of the timing-only builds
# (no gathers / no summing phase: wrong results), and the sweep kernel's phase clock (STAMPS build)
one() { name=$1; shift; env "$@" python3 tools/shard_budget.py --ranks 8 2>gpurun_out/r4/shard_$name.err | grep "G=8 rank 0" | sed "s/^measured G=8 rank 0/$name/"; }
mkdir -p gpurun_out/r4
one base X=1
for v in NOGATHER NOPHASE2; do one $v ABFT_HIP_LIB=$PWD/variants/lib_$v.so; done
one STAMPS ABFT_HIP_LIB=$PWD/variants/lib_STAMPS.so ABFT_HIP_SWEEP_DEBUG=1
grep "sweep p\|sweep w\|per XCD" gpurun_out/r4/shard_STAMPS.err | head -20
for r in 2 4 8; do for lag in 0 2; do one rpt${r}_lag$lag ABFT_HIP_SWEEP_RPT=$r ABFT_HIP_SWEEP_LAG=$lag; done; done
