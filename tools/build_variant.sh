#!/bin/bash
# Builds an experimental variant of the library: tools/build_variant.sh NAME [-DFLAG ...]
# -> build_exp/libpqd_NAME.so (select it at run time with PQD_LIBRARY=build_exp/libpqd_NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" \
  -o build_exp/libpqd_$name.so parquet-go_amd/csrc/*.hip
