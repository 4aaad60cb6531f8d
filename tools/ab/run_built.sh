#!/bin/bash
# Time the libraries built by tools/ab/build_local.sh on the GPU box, interleaved, twice:
#   gpurun -- tools/ab/run_built.sh name1 name2 ...     (AB_CONFIG=c2 AB_SPP=0 AB_REAL=f64 AB_ORDER=auto AB_VARIANT=0)
cd "$GRAFT_REPO_ROOT"
CFG=${AB_CONFIG:-c2}; SPP=${AB_SPP:-0}; REAL=${AB_REAL:-f64}; ORDER=${AB_ORDER:-auto}; VARIANT=${AB_VARIANT:-0}
for i in 1 2; do
  for n in "$@"; do RTK_DEV_TOOLS=1 RTK_HIP_LIB=$PWD/tools/ab/build/$n.so python3 tools/render_once.py $CFG $REAL 3 $SPP $VARIANT $ORDER | tail -2 | cut -c1-48 | tr "\n" " " | sed "s/^/$n ($CFG $ORDER): /"; echo; done
done
