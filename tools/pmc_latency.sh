#!/bin/bash
# Latency-type PMC passes (Little's law: mean latency = LEVEL / INSTS).  tools/pmc_latency.sh <outdir> [render_once args]
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
ARGS="$*"
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 tools/render_once.py $ARGS > "$out/$name.log" 2>&1 || echo "pass $name failed"; }
pass lat1 SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES
pass lat2 SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VSKIPPED
python3 tools/pmc_summary.py "$out" > "$out/summary.txt" 2>&1 || true
cat "$out/summary.txt"
