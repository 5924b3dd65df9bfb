#!/bin/bash
# Collects the round-3 rocprofv3 evidence on the GPU box (run via gpurun from the repo root):
# per workload one kernel trace (+ the bench line of that run) and separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / TCC hit-miss), as the MI355X guide prescribes.  Raw CSVs stay in
# gpurun_out/; the summaries written next to them are what gets copied into profiles/r03/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r03
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
# config 5's two launches separately (where the traffic goes: DESIGN.md section 4, "Round 3")
ROUND=prof_r03 tools/pmc_phases.sh c5 spmv_coo_panels 2 -- --fmt coo --mode sec7 --spec powerlaw:2097152,2 > $O/pmc_phases_powerlaw_coo_sec7.txt 2>&1
