#!/bin/bash
# PMC passes over the BENCH COMMAND ITSELF for one config (run on the GPU box through gpurun), summarised into
# profiles/<round>_pmc_<config>.json -- the file bench.py's roofline reads (keyed by kernel name, workload and the
# hash of the kernel sources; bench.py refuses a stale one).  Counters are collected in their own runs, never together
# with --kernel-trace/--stats; the program goes directly after `--` (no env/bash hop under rocprofv3).
#   tools/pmc_collect.sh <config> [round=r02] [steps=2]
set -e
cfg=$1; round=${2:-r03}; steps=${3:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_${round}_${cfg}
rm -rf "$out"; mkdir -p "$out" profiles
python3 -c "import bench; print(bench.kernel_source_hash())" > "$out/source_hash.txt"   # the sources the profiled library was built from
BENCH="bench.py --gpus 1 --config $cfg --steps $steps --warmup 1 --no-cpu-baseline --no-f32 --no-other-order --no-other-configs --no-microbench"
pass() { name=$1; shift; echo "pass $name ($cfg)"; rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 $BENCH > "$out/$name.log" 2>&1 || echo "pass $name failed"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pass sq2 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES
pass sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass sq4 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY SQ_INSTS_FLAT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
pass tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
pass tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# kernel durations of the same command (own run: --kernel-trace --stats only)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 $BENCH > "$out/stats.log" 2>&1 || echo "stats pass failed"
python3 tools/pmc_summary.py "$out" "$cfg" "$round" | tee "$out/summary.txt"
