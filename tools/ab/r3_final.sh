#!/bin/bash
# round 3, final GPU call: the GPU suite, the default bench line, then PMC + kernel-stats collection for every config.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_final_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_final_tests.log | tail -8
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python3 bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; rc=$?; echo "bench rc=$rc"; [ $rc -eq 0 ] || { tail -5 gpurun_out/r03_bench_default.err; exit 1; }
for cfg in c2 c3 c4 c5; do
  timeout -k 10 1500 tools/pmc_collect.sh $cfg r03 2 > gpurun_out/r03_pmc_collect_$cfg.log 2>&1 || { echo "pmc $cfg failed"; tail -5 gpurun_out/r03_pmc_collect_$cfg.log; exit 1; }
  tail -3 gpurun_out/r03_pmc_collect_$cfg.log | cut -c1-200
  cp profiles/r03_pmc_$cfg.json profiles/r03_kernel_stats_$cfg.csv gpurun_out/ 2>/dev/null
  timeout -k 10 600 python3 bench.py --config $cfg --no-cpu-baseline --no-other-configs > gpurun_out/r03_bench_$cfg.log 2>/dev/null || { echo "bench $cfg failed"; exit 1; }
done
echo ALL DONE
