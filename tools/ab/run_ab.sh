#!/bin/bash
# A/B several versions of rtk_trace.hip on the same MI355X box (hipcc codegen is sensitive to small source changes):
#   put candidate sources at tools/ab/<name>.hip, then gpurun -- tools/ab/run_ab.sh name1 name2 ...
# Each is built into its own library and timed on C2 next to the committed kernel (RTK_HIP_LIB override).
cd "$GRAFT_REPO_ROOT"
C=raytracingoneweekendapplication_amd/csrc
for n in "$@"; do
  mkdir -p gpurun_out/ab/$n
  cp tools/ab/$n.hip gpurun_out/ab/$n/rtk_trace.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$C $C/rtk_api.cpp $C/rtk_multi.cpp $C/rtk_optimize.cpp gpurun_out/ab/$n/rtk_trace.hip -o gpurun_out/ab/$n.so &
done
wait
CFG=${AB_CONFIG:-c2}; SPP=${AB_SPP:-0}; REAL=${AB_REAL:-f64}
for i in 1 2; do
  for n in "$@"; do RTK_DEV_TOOLS=1 RTK_HIP_LIB=$PWD/gpurun_out/ab/$n.so python tools/render_once.py $CFG $REAL 3 $SPP | tail -1 | cut -c1-48 | sed "s/^/$n ($CFG): /"; done
  python tools/render_once.py $CFG $REAL 3 $SPP | tail -1 | cut -c1-48 | sed "s/^/current ($CFG): /"
done
