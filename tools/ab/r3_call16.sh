#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
AB_CONFIG=c4 AB_SPP=0 timeout -k 10 500 tools/ab/run_built.sh m768 m1024 cold1024 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee gpurun_out/r3_call16_ab.log
