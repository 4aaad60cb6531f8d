#!/bin/bash
# PMC passes for the render kernel (run on the GPU box through gpurun).  Counters are
# collected in their own runs, never together with --kernel-trace/--stats.
#   tools/pmc_profile.sh <outdir> [render_once.py args...]
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 tools/render_once.py $ARGS > "$out/$name.log" 2>&1 || echo "pass $name failed"; }
ARGS="$*"
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pass sq2 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES
pass sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
pass tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py "$out" > "$out/summary.txt" 2>&1 || true
cat "$out/summary.txt"
