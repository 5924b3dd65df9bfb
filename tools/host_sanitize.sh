#!/bin/bash
# The host-side arithmetic (partition planner, outbox layout, generators, Matrix-Market loader) under
# AddressSanitizer + UBSan on the CPU: builds libabft_host.so with the sanitizers into a scratch
# directory, swaps it in for the run of the CPU tests that load it, and puts the regular build back.
# (GPU sanitizers are not available on the pool; this covers the code that decides shapes and offsets.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
H=$ROOT/abft_sparse_cg_amd/host
T=$(mktemp -d)
g++ -std=c++11 -O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $T/libabft_host.so \
    $H/generators.cpp $H/matrix_io.cpp $H/partition.cpp
cp $ROOT/abft_sparse_cg_amd/libabft_host.so $T/orig.so
trap 'cp $T/orig.so $ROOT/abft_sparse_cg_amd/libabft_host.so; rm -rf $T' EXIT
cp $T/libabft_host.so $ROOT/abft_sparse_cg_amd/libabft_host.so
cd $ROOT
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests/test_partition.py tests/test_host_logic.py -x -q -m "not gpu" -k "not gloo and not world"
# the TCP layer between the ranks (host/comm.cpp: rendezvous, bcast, all-reduce, all-gather(v), exchange,
# barrier) the same way, with leak detection, at 2, 3 and 5 processes
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Wall -I $H $H/comm_test.cpp $H/comm.cpp \
    $H/comm_rccl.cpp -o $T/comm_test
for n in 2 3 5; do
  P=$((21000 + RANDOM % 5000)); pids=""
  for r in $(seq 0 $((n - 1))); do
    WORLD_SIZE=$n RANK=$r LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=$P timeout 60 $T/comm_test > $T/out_$r.txt 2>&1 &
    pids="$pids $!"
  done
  st=0; for p in $pids; do wait $p || st=1; done
  echo "comm_test under ASan/UBSan, $n ranks: status $st, rank 0 says: $(tail -1 $T/out_0.txt)"
  [ $st -eq 0 ] || { cat $T/out_*.txt; exit 1; }
done
# the CPU oracle (oracle/abft_oracle.c: test infrastructure) against the golden vectors and the reference build
O=$ROOT/oracle
cp $O/libabft_oracle.so $T/oracle_orig.so
trap 'cp $T/orig.so $ROOT/abft_sparse_cg_amd/libabft_host.so; cp $T/oracle_orig.so $O/libabft_oracle.so; rm -rf $T' EXIT
gcc -O1 -g -ffp-contract=off -fopenmp -fPIC -Wall -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o $O/libabft_oracle.so $O/abft_oracle.c
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests/test_oracle_golden.py tests/test_oracle_vs_ref.py -x -q -m "not gpu"
