#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in c3 c5 c4 c2; do
  spp=0; [ $cfg = c3 ] && spp=100; [ $cfg = c5 ] && spp=32; [ $cfg = c4 ] && spp=64
  for pc in 0 1.0 1.5 2.0 2.5 3.0 4.0; do
    AB_PRIM_COST=$pc timeout -k 10 300 python3 tools/render_once.py $cfg f64 3 $spp 0 auto 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-60 | tr "\n" " " | sed "s/^/prim_cost=$pc ($cfg): /"; echo
  done
done | tee gpurun_out/r3_primcost.log
