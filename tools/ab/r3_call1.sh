#!/bin/bash
# round 3, GPU call 1: parity suite on the new default build, then the C5 A/B of the deferred texture step, the rare-record
# loop and the medium's certain-miss test, then the C5 phase profile.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_call1_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3_call1_tests.log
tail -3 gpurun_out/r3_call1_tests.log
AB_CONFIG=c5 AB_SPP=32 timeout -k 10 600 tools/ab/run_built.sh c5_base c5_defer c5_loop c5_all c5_all768 2>&1 | tee gpurun_out/r3_call1_ab_c5.log
RTK_PROF_LIB=$PWD/tools/ab/build/c5_prof.so timeout -k 10 300 python3 tools/profile_phases.py c5 f64 32 2>&1 | tee gpurun_out/r3_call1_phases_c5.log
