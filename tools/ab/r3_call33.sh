#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 300 python3 -c "
import json, raytracingoneweekendapplication_amd as rt
for k in range(2):
    print(json.dumps(rt.microbench(0), indent=1))
" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_call33.log
