#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
AB_CONFIG=c5 AB_SPP=32 timeout -k 10 500 tools/ab/run_built.sh now pow log pow_log pow_1024 pow_log_1024 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee gpurun_out/r3_call12_ab.log
