#!/bin/bash
# diagnostic copy of the library with s_memtime stamps in token_block_kernel (TB_STAMP): scratch/stamp/libsam6d_hip.so
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/stamp
C=openvino-sam-6d_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fno-vectorize -DTB_STAMP -c $C/block.hip -o scratch/stamp/block.o
objs=$(ls $C/*.o | grep -v "/block.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/stamp/libsam6d_hip.so scratch/stamp/block.o $objs
echo built scratch/stamp/libsam6d_hip.so
