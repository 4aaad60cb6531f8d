#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
AB_CONFIG=c2 AB_SPP=0 timeout -k 10 400 tools/ab/run_built.sh pk0 pk1 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee -a gpurun_out/r3_call6_ab.log
