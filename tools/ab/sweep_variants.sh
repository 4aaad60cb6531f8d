#!/bin/bash
# Time one built library (tools/ab/build/<name>.so) under a list of variant words:
#   gpurun -- tools/ab/sweep_variants.sh <name> v1 v2 ...     (AB_CONFIG=c2)
cd "$GRAFT_REPO_ROOT"
n=$1; shift
for v in "$@"; do RTK_DEV_TOOLS=1 RTK_HIP_LIB=$PWD/tools/ab/build/$n.so python3 tools/render_once.py ${AB_CONFIG:-c2} f64 3 0 $v auto 2>&1 | tail -2 | cut -c1-40 | tr "\n" " " | sed "s/^/$n variant $v: /"; echo; done
