#!/bin/bash
# second half of the round-3 collection (a gpurun call is at most 20 minutes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r03
mkdir -p $O
# the reference's benchmark method on config 2's matrix and on config 4's (constraints on the sweep layout)
NUM_RUNS=3 abft_sparse_cg_amd/host/run_benchmark abft_sparse_cg_amd/host/cg-csr -s laplace5:3162,3162 -c 0 -i 200 > $O/run_benchmark_csr.txt 2>&1
NUM_RUNS=3 abft_sparse_cg_amd/host/run_benchmark abft_sparse_cg_amd/host/cg-coo -s laplace5:3162,3162 -c 0 -i 200 > $O/run_benchmark_coo.txt 2>&1
NUM_RUNS=3 abft_sparse_cg_amd/host/run_benchmark abft_sparse_cg_amd/host/cg-csr -s random:4194304,24,1 -c 0 -i 100 > $O/run_benchmark_csr_random.txt 2>&1
python3 tools/shard_budget.py --spec random:4194304,24,1 --mode secded --ranks 1,2,4,8 > $O/shard_budget_config4.md 2> $O/shard_budget_config4.err
ls $O

echo done; ls $O
