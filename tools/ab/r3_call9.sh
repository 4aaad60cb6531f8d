#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
AB_CONFIG=c5 AB_SPP=32 timeout -k 10 500 tools/ab/run_built.sh base colduv coldboth coldboth768 colduv768 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee gpurun_out/r3_call9_ab.log
