#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
AB_CONFIG=c5 AB_SPP=32 timeout -k 10 400 tools/ab/run_built.sh nopair pair 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee gpurun_out/r3_call7_ab.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_tie_break.py -m gpu -q -x -k "not stated_sizes or c5" > gpurun_out/r3_call7_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r3_call7_tests.log | tail -12
