#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python3 tools/knockout.py 32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_call32.log
