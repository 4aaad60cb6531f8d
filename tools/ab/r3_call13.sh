#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
AB_CONFIG=c5 AB_SPP=32 timeout -k 10 500 tools/ab/run_built.sh pl pl_perlin pl_noqpf pl_nochpf pl_512 pl_unr2 pl_unr1 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee gpurun_out/r3_call13_ab.log
