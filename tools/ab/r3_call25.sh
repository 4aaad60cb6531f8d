#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_call25_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r3_call25_tests.log | tail -12
export RTK_DEV_TOOLS=1
AB_CONFIG=c5 AB_SPP=32 timeout -k 10 300 tools/ab/run_built.sh nosmo smo 2>&1 | grep -v amdgpu.ids | cut -c1-120
for n in nosmo smo; do
  export RTK_HIP_LIB=$PWD/tools/ab/build/$n.so
  rm -rf gpurun_out/pmc_$n; timeout -k 10 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_$n -- python3 tools/render_once.py c5 f64 2 32 0 auto > gpurun_out/pmc_$n.log 2>&1 || { echo "rocprofv3 failed for $n"; exit 1; }
  python3 - <<PY
import csv,glob
rows=[r for f in glob.glob('gpurun_out/pmc_$n/**/*counter_collection.csv',recursive=True) for r in csv.DictReader(open(f)) if 'rtk_render_kernel' in r['Kernel_Name']]
by={}
for r in rows: by.setdefault((r['Dispatch_Id'],r['Counter_Name']),0.0); by[(r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value'])
v=[x for (d,k),x in by.items() if k=='WRITE_SIZE']; print('$n WRITE_SIZE GB per dispatch', sum(v)/max(len(v),1)*1024/1e9, 'dispatches',len(v))
PY
done
