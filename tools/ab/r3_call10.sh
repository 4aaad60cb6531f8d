#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
AB_CONFIG=c5 AB_SPP=32 timeout -k 10 500 tools/ab/run_built.sh base l0_768 l1_768 l2_768 l2_1024 l2_512 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee gpurun_out/r3_call10_ab.log
