#!/bin/bash
# Build A/B variants of the kernel library HERE (hipcc cross-compiles gfx950 without a GPU); the .so files travel
# to the GPU box with the snapshot (tools/ab/build/*.so is git-ignored, not gpurun-ignored).
#   tools/ab/build_local.sh name1:"-DRTK_AB_X" name2:"-DRTK_AB_Y -DRTK_AB_Z" ...
cd "$(dirname "$0")/../.."
C=raytracingoneweekendapplication_amd/csrc
mkdir -p tools/ab/build
for spec in "$@"; do
  n=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $flags -Iinclude -I$C $C/rtk_api.cpp $C/rtk_multi.cpp $C/rtk_optimize.cpp $C/rtk_trace.hip -o tools/ab/build/$n.so &
done
wait
ls -la tools/ab/build
