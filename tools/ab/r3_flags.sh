#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in c2 c4 c3 c5; do
  spp=0; [ $cfg = c3 ] && spp=100; [ $cfg = c5 ] && spp=32; [ $cfg = c4 ] && spp=64
  AB_CONFIG=$cfg AB_SPP=$spp timeout -k 10 600 tools/ab/run_built.sh base maxilp memclause nomisched O2 nopostra 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3_flags.log
