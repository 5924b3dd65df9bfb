set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r01
mkdir -p $O
rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- python3 bench.py --cpu-iters 0 --steps 100 > $O/bench_under_trace.json 2> $O/trace.err
python3 profiles/summarize.py trace $O/trace $O/kernel_trace_laplace_none.md
rocprofv3 --pmc FETCH_SIZE -f csv -d $O/pmc_fetch -- python3 bench.py --cpu-iters 0 --steps 10 --warmup 2 --no-profile > /dev/null 2> $O/pmc_fetch.err
python3 profiles/summarize.py pmc $O/pmc_fetch $O/pmc_fetch_laplace_none.json
rocprofv3 --pmc WRITE_SIZE -f csv -d $O/pmc_write -- python3 bench.py --cpu-iters 0 --steps 10 --warmup 2 --no-profile > /dev/null 2> $O/pmc_write.err
python3 profiles/summarize.py pmc $O/pmc_write $O/pmc_write_laplace_none.json
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -f csv -d $O/pmc_tcc -- python3 bench.py --cpu-iters 0 --steps 10 --warmup 2 --no-profile > /dev/null 2> $O/pmc_tcc.err
python3 profiles/summarize.py pmc $O/pmc_tcc $O/pmc_tcc_laplace_none.json
rocprofv3 --pmc FETCH_SIZE -f csv -d $O/pmc_fetch_rand -- python3 bench.py --cpu-iters 0 --steps 5 --warmup 1 --no-profile --mode secded --spec random:4194304,24,1 > /dev/null 2> $O/pmc_fetch_rand.err
python3 profiles/summarize.py pmc $O/pmc_fetch_rand $O/pmc_fetch_random_secded.json
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -f csv -d $O/pmc_tcc_rand -- python3 bench.py --cpu-iters 0 --steps 5 --warmup 1 --no-profile --mode secded --spec random:4194304,24,1 > /dev/null 2> $O/pmc_tcc_rand.err
python3 profiles/summarize.py pmc $O/pmc_tcc_rand $O/pmc_tcc_random_secded.json
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/pmc_fetch_rand $O/pmc_tcc_rand
ls -la $O
