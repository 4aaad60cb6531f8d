#!/bin/bash
# round 3, GPU call 2: the whole GPU suite on the new build (full-size configs, device KATs, async multi path), the measured
# ceilings, then the rare-record loop A/B on C5 / C3 / C4.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r3_call2_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3_call2_tests.log
tail -15 gpurun_out/r3_call2_tests.log
timeout -k 10 120 python3 -c "
import raytracingoneweekendapplication_amd as rt, json
print(json.dumps(rt.microbench(0), indent=1))" 2>&1 | tee gpurun_out/r3_call2_microbench.log
for cfg in c5:32 c3:100 c4:64 c2:0; do
  AB_CONFIG=${cfg%%:*} AB_SPP=${cfg##*:} timeout -k 10 400 tools/ab/run_built.sh loop0 loop2 loop4 current 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3_call2_ab.log
done
