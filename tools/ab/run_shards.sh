#!/bin/bash
# Like run_built.sh, but times every rank's shard for N = 1 and N = 8 (tools/rank_probe.py) with each library:
#   gpurun -- tools/ab/run_shards.sh name1 name2 ...
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for n in "$@"; do RTK_DEV_TOOLS=1 RTK_HIP_LIB=$PWD/tools/ab/build/$n.so python3 tools/rank_probe.py ${AB_RANKS:-1,8} ${AB_SPP:-0} | cut -c1-150 | sed "s/^/$n: /"; done
done
