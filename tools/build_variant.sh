#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...] : an A/B build of the library as variants/lib_NAME.so (ABFT_HIP_LIB picks it up)
N=$1; shift
mkdir -p variants
cd abft_sparse_cg_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall \
  -Wno-unused-function -shared "$@" -o ../../variants/lib_$N.so kernels.hip abft_hip.hip
