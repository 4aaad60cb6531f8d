#!/bin/bash
# round 3: refresh of the bench lines on unchanged kernel sources (the PMC profiles stay valid): GPU suite, default line, one line per config.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_final_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_final_tests.log | tail -8
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python3 bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; rc=$?; echo "bench rc=$rc"; [ $rc -eq 0 ] || { tail -5 gpurun_out/r03_bench_default.err; exit 1; }
for cfg in c2 c3 c4 c5; do
  timeout -k 10 600 python3 bench.py --config $cfg --no-cpu-baseline --no-other-configs > gpurun_out/r03_bench_$cfg.log 2>/dev/null || { echo "bench $cfg failed"; exit 1; }
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r03_bench_$cfg.log') if l.startswith('{')][-1]); r=d['roofline']
print('$cfg', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'of measured', (r.get('frac_of_measured_issue_ceiling') or {}).get('value'), 'lane', r.get('valu_lane_utilisation'), 'work', r.get('work_frac'), d['framebuffer_sha256'])"
done
echo ALL DONE
