#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2; do
for cfg in c2 c3 c4 c5; do
  spp=0; [ $cfg = c3 ] && spp=100; [ $cfg = c5 ] && spp=32; [ $cfg = c4 ] && spp=64
  for fl in 0 1; do
    RTK_OPT_FLATTEN=$fl timeout -k 10 300 python3 tools/render_once.py $cfg f64 3 $spp 0 auto 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-60 | tr "\n" " " | sed "s/^/flatten=$fl ($cfg): /"; echo
  done
done
done | tee gpurun_out/r3_flatten.log
