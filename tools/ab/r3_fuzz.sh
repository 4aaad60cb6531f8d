#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "zoo or big or importing_torch or degenerate or sphere_scenes" 2>&1 | tail -8
timeout -k 10 1000 python3 tools/fuzz_parity.py ${FUZZ_N:-150} 50000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_fuzz_parity.txt | tail -14
