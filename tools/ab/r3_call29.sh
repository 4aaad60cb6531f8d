#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_call29_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r3_call29_tests.log | tail -12
for cfg in c3:200 c4:0 c5:32; do
  AB_CONFIG=${cfg%%:*} AB_SPP=${cfg##*:} timeout -k 10 400 tools/ab/run_built.sh old new 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee -a gpurun_out/r3_call29_ab.log
done
