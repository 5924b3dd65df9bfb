#!/bin/bash
# config 5: spmv_coo_lean_kernel (ABFT_HIP_COO_LEAN=1) by outputs per thread (variant builds), panel width and pacing lag
run() { env "$@" timeout -k 5 120 python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 40 --warmup 4 --fmt coo --mode ${MODE:-sec7} --spec powerlaw:2097152,2 2>&1 | grep -o 'avg_us": [0-9.]*\|hip:.*' | head -2 | tr '\n' ' '; echo; }
echo "one-role kernel, default:"; run X=1
for v in L8W5 L8W6; do echo "== $v (8 outputs per thread)"; for lag in 0 2; do echo -n "lag $lag: "; run ABFT_HIP_LIB=$PWD/variants/lib_$v.so ABFT_HIP_COO_LEAN=1 ABFT_HIP_PANEL_LAG=$lag; done; done
echo "== L4 (4 outputs per thread, 8 workgroups per CU)"
for w in 155000 262144 320000; do for lag in 0 2 3; do echo -n "width $w lag $lag: "; run ABFT_HIP_LIB=$PWD/variants/lib_L4.so ABFT_HIP_COO_LEAN=1 ABFT_HIP_PANEL_LAG=$lag ABFT_HIP_PANEL_WIDTH=$w; done; done
