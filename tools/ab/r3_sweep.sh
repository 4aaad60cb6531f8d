#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in c3 c5; do
  spp=100; [ $cfg = c5 ] && spp=32
  for v in 0 256 512 768 1024 1280 1536 16384 32768 49152 65536; do
    timeout -k 10 300 python3 tools/render_once.py $cfg f64 3 $spp $v auto 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-48 | tr "\n" " " | sed "s/^/$cfg variant $v: /"; echo
  done
done | tee gpurun_out/r3_sweep.log
