#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python3 bench.py --gpus 2 --rehearse-one-gpu --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>gpurun_out/r3_call30.err | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rehearse N=2', d['value'], d['ms_per_step'], d['framebuffer_sha256']); print(json.dumps(d['multi_paths'])[:900])"
tail -3 gpurun_out/r3_call30.err | cut -c1-200
