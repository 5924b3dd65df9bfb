#!/bin/bash
# SpMV time of the same binary over several fresh processes on one box (placement luck?)
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --cpu-iters 0 --no-probe --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('run $i: %.1f it/s  SpMV %.1f us' % (d['value'], d['roofline']['avg_launch_us']))"
done
