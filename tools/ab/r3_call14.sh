#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python3 tools/variant_sweep.py c5 auto f64 0,256,768,1024,16384,49152,65536,393216,524288,655360,786432,131072 32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_call14_sweep_c5.log
