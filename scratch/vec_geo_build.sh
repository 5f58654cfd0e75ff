#!/bin/bash
# diagnostic copy of the library with geo.hip compiled WITH the SLP / loop vectorisers (packed-fp32 VALU instructions in
# geo_index_kernel), everything else as shipped: scratch/vecgeo/libsam6d_hip.so
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/vecgeo
C=openvino-sam-6d_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -c $C/geo.hip -o scratch/vecgeo/geo.o
objs=$(ls $C/*.o | grep -v "/geo.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/vecgeo/libsam6d_hip.so scratch/vecgeo/geo.o $objs
echo built scratch/vecgeo/libsam6d_hip.so
