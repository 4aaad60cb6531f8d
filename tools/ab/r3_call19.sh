#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in c2:0 c3:200 c4:0 c5:32; do
  c=${cfg%%:*}; spp=${cfg##*:}
  RTK_PROF_LIB=$PWD/tools/ab/build/prof.so timeout -k 10 300 python3 tools/profile_phases.py $c f64 $spp 2>&1 | grep -v amdgpu.ids | head -14 | cut -c1-230 > gpurun_out/r03_phases_$c.txt
  cat gpurun_out/r03_phases_$c.txt | head -13
done
