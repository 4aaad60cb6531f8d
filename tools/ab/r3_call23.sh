#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for ml in 0 6 8 12 16 24; do for pc in 0 0.5 0.25 0.1; do
  AB_MAX_LEAF=$ml AB_PRIM_COST=$pc python3 tools/render_once.py c3 f64 3 100 0 auto 2>&1 | tail -2 | cut -c1-56 | tr "\n" " " | sed "s/^/c3 max_leaf=$ml prim_cost=$pc: /"; echo
done; done | tee gpurun_out/r3_call23_c3_sweep.log
