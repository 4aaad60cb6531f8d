#!/bin/bash
# round 3, GPU call 5: measured issue rates incl. packed f32, then the packed box step A/B on C2 (+ shards)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 120 python3 -c "
import raytracingoneweekendapplication_amd as rt, json
print(json.dumps(rt.microbench(0), indent=1))" 2>&1 | tee gpurun_out/r3_call5_microbench.log
AB_CONFIG=c2 AB_SPP=0 timeout -k 10 400 tools/ab/run_built.sh pk0 pk1 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee -a gpurun_out/r3_call5_ab.log
