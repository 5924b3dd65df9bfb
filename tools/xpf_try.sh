#!/bin/bash
# config 5: the x prefetch of the COO panel kernel (ABFT_HIP_PANEL_XPF) by panel width and pacing lag
run() { env "$@" timeout -k 5 120 python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 40 --warmup 4 --fmt coo --mode ${MODE:-sec7} --spec ${SPEC:-powerlaw:2097152,2} 2>&1 | grep -o 'avg_us": [0-9.]*\|hip:.*' | head -2 | tr '\n' ' '; echo; }
for w in 262144 200000 163840 131072; do for lag in 2 3; do for x in 0 1; do echo -n "width $w lag $lag xpf $x: "; run ABFT_HIP_PANEL_LAG=$lag ABFT_HIP_PANEL_WIDTH=$w ABFT_HIP_PANEL_XPF=$x $EXTRA; done; done; done
