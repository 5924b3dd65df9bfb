#!/bin/bash
# A/B of library builds under variants/ (scratch): one bench.py process per variant.
# usage: tools/ab_variants.sh "<bench args>" variant...
ARGS=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset ABFT_HIP_LIB; else export ABFT_HIP_LIB=$PWD/variants/lib_$v.so; fi
  python bench.py --cpu-iters 0 --profile-all $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('%-12s it/s %8.1f  ' % ('$v', d['value']) + '  '.join('%s %.1fus' % (n, k[n]['avg_us']) for n in k))"
done
