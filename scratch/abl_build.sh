#!/bin/bash
# timing-only copies of the library with one ingredient of token_block_kernel's panel loop removed (results are garbage):
# scratch/abl/<noread|nodma|nobar|noreaddma>/libsam6d_hip.so
set -e
cd "$(dirname "$0")/.."
C=openvino-sam-6d_amd/csrc
objs=$(ls $C/*.o | grep -v "/block.o")
for v in "noread -DTB_ABL_NOREAD" "nodma -DTB_ABL_NODMA" "nobar -DTB_ABL_NOBAR" "noreaddma -DTB_ABL_NOREAD -DTB_ABL_NODMA"; do
  set -- $v; name=$1; shift
  mkdir -p scratch/abl/$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fno-vectorize "$@" -c $C/block.hip -o scratch/abl/$name/block.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/abl/$name/libsam6d_hip.so scratch/abl/$name/block.o $objs
  rm scratch/abl/$name/block.o
done
echo built scratch/abl
