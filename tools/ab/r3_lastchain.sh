#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_final_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_final_tests.log | tail -8
for i in 1 2; do
for cfg in c3 c5; do
  spp=100; [ $cfg = c5 ] && spp=32
  for keep in 1 0; do
    if [ $keep = 1 ]; then export RTK_KEEP_LAST_CHAIN=1; else unset RTK_KEEP_LAST_CHAIN; fi
    timeout -k 10 300 python3 tools/render_once.py $cfg f64 3 $spp 0 auto 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-60 | tr "\n" " " | sed "s/^/keep_last_chain=$keep ($cfg): /"; echo
  done
done
done
unset RTK_KEEP_LAST_CHAIN
timeout -k 10 600 python3 tools/fuzz_parity.py 300 60000 2>&1 | grep -v amdgpu.ids | tail -4
