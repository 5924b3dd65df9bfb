#!/bin/bash
# second half of the round-4 collection (a gpurun call is at most 20 minutes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r04
mkdir -p $O
python3 tools/shard_budget.py --spec random:4194304,24,1 --mode secded --ranks 1,2,4,8 > $O/shard_budget_config4.md 2> $O/shard_budget_config4.err
ABFT_CG_TAIL=0 python3 tools/shard_budget.py --spec random:4194304,24,1 --mode secded --ranks 8 > $O/shard_budget_config4_three_kernels.md 2> $O/shard_budget_config4_three_kernels.err
python3 tools/shard_budget.py --spec laplace5:3162,3162 --mode none --ranks 1,2,4,8 > $O/shard_budget_config2.md 2> $O/shard_budget_config2.err
NUM_RUNS=3 abft_sparse_cg_amd/host/run_benchmark abft_sparse_cg_amd/host/cg-csr -s laplace5:3162,3162 -c 0 -i 200 > $O/run_benchmark_csr.txt 2>&1
NUM_RUNS=3 abft_sparse_cg_amd/host/run_benchmark abft_sparse_cg_amd/host/cg-coo -s powerlaw:2097152,2 -c 0 -i 100 > $O/run_benchmark_coo_powerlaw.txt 2>&1
tools/tail_ab.sh > $O/tail_ab.txt 2>&1
ls $O
