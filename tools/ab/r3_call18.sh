#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in 0 1 1048576 2097152 2097153; do AB_CONFIG=c5 AB_SPP=16 AB_VARIANT=$v timeout -k 10 300 tools/ab/run_built.sh a512 new 2>&1 | grep -v amdgpu.ids | cut -c1-100 | sed "s/^/v$v /" | tee -a gpurun_out/r3_call18_ab.log; done
AB_CONFIG=c4 AB_SPP=64 AB_VARIANT=1 timeout -k 10 300 tools/ab/run_built.sh new 2>&1 | grep -v amdgpu.ids | cut -c1-100 | tee -a gpurun_out/r3_call18_ab.log
AB_CONFIG=c4 AB_SPP=64 timeout -k 10 300 tools/ab/run_built.sh new 2>&1 | grep -v amdgpu.ids | cut -c1-100 | tee -a gpurun_out/r3_call18_ab.log
