#!/bin/bash
# Collects the round-1 rocprofv3 evidence on the GPU box (run via gpurun from the repo root):
#   kernel trace + stats of the default bench.py workload, then separate PMC passes
#   (FETCH_SIZE / WRITE_SIZE / TCC hit-miss) as the MI355X guide prescribes, for the
#   Laplacian (config 2) and the random matrix (config 4).  Raw CSVs stay in gpurun_out/;
#   the summaries written next to them are what gets copied into profiles/r01/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r01
rm -rf $O; mkdir -p $O
B="python3 bench.py --cpu-iters 0"
rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- $B --steps 100 > $O/bench_under_trace.json 2> $O/trace.err
python3 profiles/summarize.py trace $O/trace $O/kernel_trace_laplace_none.md
rocprofv3 --kernel-trace --stats -f csv -d $O/trace_secded -- $B --steps 100 --mode secded > $O/bench_under_trace_secded.json 2> $O/trace2.err
python3 profiles/summarize.py trace $O/trace_secded $O/kernel_trace_laplace_secded.md
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $c | cut -d' ' -f1 | tr A-Z a-z)
  rocprofv3 --pmc $c -f csv -d $O/pmc_$n -- $B --steps 10 --warmup 2 --no-profile > /dev/null 2> $O/pmc_$n.err
  python3 profiles/summarize.py pmc $O/pmc_$n $O/pmc_${n}_laplace_none.json > /dev/null
  rocprofv3 --pmc $c -f csv -d $O/pmcr_$n -- $B --steps 5 --warmup 1 --no-profile --mode secded --spec random:4194304,24,1 > /dev/null 2> $O/pmcr_$n.err
  python3 profiles/summarize.py pmc $O/pmcr_$n $O/pmc_${n}_random_secded.json > /dev/null
done
rocprofv3 --kernel-trace --stats -f csv -d $O/trace_rand -- $B --steps 30 --mode secded --spec random:4194304,24,1 > $O/bench_under_trace_random_secded.json 2> $O/trace3.err
python3 profiles/summarize.py trace $O/trace_rand $O/kernel_trace_random_secded.md
rocprofv3 --kernel-trace --stats -f csv -d $O/trace_coo -- $B --steps 50 --fmt coo --mode sec7 --spec powerlaw:2097152,2 > $O/bench_under_trace_coo_sec7.json 2> $O/trace4.err
python3 profiles/summarize.py trace $O/trace_coo $O/kernel_trace_powerlaw_coo_sec7.md
rm -rf $O/trace_rand $O/trace_coo
rm -rf $O/trace $O/trace_secded $O/pmc_fetch_size $O/pmc_write_size $O/pmc_tcc_hit_sum $O/pmcr_fetch_size $O/pmcr_write_size $O/pmcr_tcc_hit_sum
cat $O/kernel_trace_laplace_none.md
ls $O
