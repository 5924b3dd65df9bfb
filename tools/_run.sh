mkdir -p gpurun_out/r3
C5="--spec powerlaw:2097152,2 --fmt coo --mode sec7"
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 5 120 python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 40 --warmup 5 $C5 "$@" 2>gpurun_out/r3/c5_$name.err | python3 -c "
import json,sys
try:
  d=json.loads(sys.stdin.read()); print('%-28s spmv %8.1f us   %7.1f it/s' % ('$name', d['kernels']['spmv']['avg_us'], d['value']))
except Exception as e: print('$name FAILED', e)"; }
run base ABFT_X=1 --
for w in 200000 220000 232000 240000 244000 248000 252000 256000 300000 331000; do run w$w ABFT_HIP_PANEL_WIDTH=$w --; done
for ch in 0 2 8; do run w248000_chunk$ch ABFT_HIP_PANEL_WIDTH=248000 ABFT_HIP_PANEL_CHUNK=$ch --; done
