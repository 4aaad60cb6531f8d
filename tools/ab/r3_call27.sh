#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_call27_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r3_call27_tests.log | tail -12
for cfg in c5; do python3 tools/render_once.py $cfg f64 3 0 0 auto 2>&1 | tail -2 | cut -c1-60 | tr "\n" " " | sed "s/^/$cfg: /"; echo; done
