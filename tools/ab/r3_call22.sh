#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
AB_CONFIG=c2 AB_SPP=0 timeout -k 10 400 tools/ab/run_built.sh w4 w5 w6 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee gpurun_out/r3_call22_ab.log
RTK_DEBUG=1 RTK_DEV_TOOLS=1 RTK_HIP_LIB=$PWD/tools/ab/build/w5.so python3 tools/render_once.py c2 f64 1 0 0 auto 2>&1 | grep "launch plan" | head -2
