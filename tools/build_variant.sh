#!/bin/bash
# Development aid: build fast_amd/csrc/libfasts_hip_<name>.so with extra compiler flags (objects in build_<name>/):
#   tools/build_variant.sh stamp -DFS_STAMP
set -e
NAME=$1; shift
cd "$(dirname "$0")/../fast_amd/csrc"
B=build_$NAME
mkdir -p $B
SRCS="fasts_capi fasts_comm fasts_kernels fasts_stage_dma fasts_stage_full fasts_lusgs fasts_stage_sa fasts_wallflux"
for s in $SRCS; do
  # only the stage kernel is recompiled unless ALL=1: the other objects come from the default build
  if [ "$s" = fasts_stage_dma ] || [ -n "$ALL" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function "$@" -c -o $B/$s.o $s.hip &
  else
    cp build/$s.o $B/$s.o
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libfasts_hip_$NAME.so $(for s in $SRCS; do echo $B/$s.o; done) -ldl
echo built libfasts_hip_$NAME.so
