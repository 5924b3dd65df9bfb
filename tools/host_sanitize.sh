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
