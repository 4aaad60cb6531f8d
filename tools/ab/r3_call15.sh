#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_call15_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3_call15_tests.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_call15_tests.log | tail -12
# multi-pass overhead: C5 and C3 at full spp, passes of 16 chunks (variant 0) vs one launch (variant bit 24)
for cfg in c5 c3; do for v in 0 16777216; do RTK_DEV_TOOLS=1 RTK_HIP_LIB=$PWD/tools/ab/build/new.so python3 tools/render_once.py $cfg f64 3 0 $v auto 2>&1 | tail -2 | cut -c1-60 | tr "\n" " " | sed "s/^/$cfg variant $v: /"; echo; done; done | tee gpurun_out/r3_call15_passes.log
timeout -k 10 600 python3 bench.py --steps 10 --warmup 2 > gpurun_out/r3_call15_bench_c2.json 2> gpurun_out/r3_call15_bench_c2.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r3_call15_bench_c2.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3_call15_bench_c2.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step']); r=d['roofline']; print({k:r.get(k) for k in ('frac','work_frac','kernel_ms','work_simd_cycles_per_sample')}); print(r['ceilings']['issue_cycles_per_wave_instruction'], r['hbm_model'].get('model_vs_lds_ceiling'))
print({k:(v['value'],v['other_order']['value'],v['other_order']['identical_framebuffer']) for k,v in d['other_configs'].items()}); print(d['cpu_baseline'])"
