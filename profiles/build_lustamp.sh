#!/bin/bash
# development build of the library with time stamps inside the LU-SGS sweeps (-DFS_LU_STAMP): libfasts_hip_lustamp.so
set -e
cd "$(dirname "$0")/../fast_amd/csrc"
make
mkdir -p build_stamp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -DFS_LU_STAMP -c -o build_stamp/fasts_lusgs.o fasts_lusgs.hip
objs=$(ls build/*.o | grep -v fasts_lusgs.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libfasts_hip_lustamp.so $objs build_stamp/fasts_lusgs.o -ldl
