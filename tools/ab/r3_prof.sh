#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in c3 c2 c4; do
  spp=0; [ $cfg = c3 ] && spp=100; [ $cfg = c4 ] && spp=64
  RTK_PROF_LIB=$PWD/tools/ab/build/prof.so timeout -k 10 300 python3 tools/profile_phases.py $cfg f64 $spp auto 2>&1 | grep -v amdgpu.ids | head -16
done | tee gpurun_out/r3_prof.log
