#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_call17_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3_call17_tests.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_call17_tests.log | tail -12
AB_CONFIG=c4 AB_SPP=64 AB_ORDER=reference timeout -k 10 300 tools/ab/run_built.sh m768 new 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee gpurun_out/r3_call17_ab.log
AB_CONFIG=c4 AB_SPP=64 AB_VARIANT=1 timeout -k 10 300 tools/ab/run_built.sh m768 new 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee -a gpurun_out/r3_call17_ab.log
AB_CONFIG=c4 AB_SPP=64 AB_VARIANT=1048576 timeout -k 10 300 tools/ab/run_built.sh m768 new 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee -a gpurun_out/r3_call17_ab.log
