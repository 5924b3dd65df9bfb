#!/bin/bash
# Collects the round-4 rocprofv3 evidence on the GPU box (run via gpurun from the repo root):
# per workload one kernel trace (+ the bench line of that run) and separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / TCC hit-miss), as the MI355X guide prescribes.  Raw CSVs stay in
# gpurun_out/; the summaries written next to them are what gets copied into profiles/r04/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r04
rm -rf $O; mkdir -p $O
B="python3 bench.py --cpu-iters 0 --no-probe --no-extras"
declare -A W
W[laplace_none]="--mode none"
W[laplace_sed]="--mode sed"
W[laplace_secded]="--mode secded"
W[random_secded]="--mode secded --spec random:4194304,24,1"
W[powerlaw_coo_sec7]="--fmt coo --mode sec7 --spec powerlaw:2097152,2"
for tag in laplace_none laplace_sed laplace_secded random_secded powerlaw_coo_sec7; do
  a=${W[$tag]}
  rocprofv3 --kernel-trace --stats -f csv -d $O/trace_$tag -- $B --steps 60 $a > $O/bench_under_trace_$tag.json 2> $O/trace_$tag.err
  python3 profiles/summarize.py trace $O/trace_$tag $O/kernel_trace_$tag.md > /dev/null
  rm -rf $O/trace_$tag
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d' ' -f1 | tr A-Z a-z)
    rocprofv3 --pmc $c -f csv -d $O/pmc_${tag}_$n -- $B --steps 6 --warmup 2 --no-profile $a > /dev/null 2> $O/pmc_${tag}_$n.err
    python3 profiles/summarize.py pmc $O/pmc_${tag}_$n $O/pmc_${n}_$tag.json > /dev/null
    rm -rf $O/pmc_${tag}_$n
  done
  echo "== $tag"; head -6 $O/kernel_trace_$tag.md
done
# the tail of the iteration in one launch (cg_tail_kernel) on configs[3]'s 1/8 shard size and the graph loop of config 2:
# kernel traces of the C++ driver's --bench loop
for spec in "random:524288,24,1 secded tail_shard 1" "random:524288,24,1 secded three_kernels_shard 0" "random:4194304,24,1 secded tail_config4 1"; do
  set -- $spec
  export ABFT_CG_TAIL=$4
  rocprofv3 --kernel-trace --stats -f csv -d $O/trace_$3 -- abft_sparse_cg_amd/host/cg-csr -t hip -m $2 -s $1 --bench 20,200,5 -q > $O/bench_under_trace_$3.txt 2> $O/trace_$3.err
  python3 profiles/summarize.py trace $O/trace_$3 $O/kernel_trace_$3.md > /dev/null
  rm -rf $O/trace_$3
done
unset ABFT_CG_TAIL
