#!/bin/bash
# round 3, GPU call 3: the whole GPU suite (no -x: every failure at once), then the rare-record loop A/B.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_call3_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3_call3_tests.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_call3_tests.log | tail -20
cat gpurun_out/full_size_sha_*.json 2>/dev/null
for cfg in c5:32 c3:100 c4:64; do
  AB_CONFIG=${cfg%%:*} AB_SPP=${cfg##*:} timeout -k 10 400 tools/ab/run_built.sh loop0 loop2 current 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee -a gpurun_out/r3_call3_ab.log
done
