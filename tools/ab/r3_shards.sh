#!/bin/bash
# per-rank shard times on one GPU for N = 1, 2, 4, 8 (what each of N GPUs would render): tools/rank_probe.py on the product library
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 300 python3 tools/rank_probe.py 1,2,4,8 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_shard_probe_c2.txt
