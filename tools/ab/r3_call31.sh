#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
AB_CONFIG=c5 AB_SPP=32 timeout -k 10 900 tools/ab/run_built.sh head w0 w8a6 w16a8 w4a3 w12a12 2>&1 | tee gpurun_out/r3_call31.log
