#!/bin/bash
# repeat the world-size-1 RCCL test to look for stalls (each run bounded by timeout)
for i in 1 2 3 4 5 6; do
  timeout -k 5 150 python -m pytest tests/test_gpu_distributed.py -x -q 2>&1 | tail -1
  echo "run $i rc=$?"
done
