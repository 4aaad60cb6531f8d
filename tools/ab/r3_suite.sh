#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_final_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_final_tests.log | tail -8
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
