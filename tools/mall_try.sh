#!/bin/bash
# does keeping the once-per-iteration vector x out of the caches (non-temporal in calc_px: variants/lib_XNT.so) let p, r, w
# live in the 256 MiB Infinity Cache across the kernels of an iteration?  bench.py's host loop, all four kernels bracketed
for spec in ${SPECS:-laplace5:3162,3162 laplace5:2800,2800 laplace5:2400,2400 laplace5:2000,2000}; do
  for v in ${VARIANTS:-base XNT base XNT}; do
    if [ "$v" = base ]; then unset ABFT_HIP_LIB; else export ABFT_HIP_LIB=$PWD/variants/lib_$v.so; fi
    echo -n "$spec $v: "
    python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 100 --warmup 10 --profile-all --spec $spec 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); k=d['kernels']
print(d['value'], 'it/s;', ' '.join('%s %.1f' % (n, k[n]['avg_us']) for n in k))"
  done
done
