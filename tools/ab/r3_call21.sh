#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
timeout -k 10 600 python3 bench.py --gpus 2 --rehearse-one-gpu --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rehearse N=2', d['value'], d['ms_per_step'], d['framebuffer_sha256'], d['config']['parallelism'])"
timeout -k 10 600 python3 bench.py --multi abi --gpus 4 --multi-devices 0,0,0,0 --steps 5 --warmup 1 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('abi 4 slots', d['value'], d['ms_per_step'], d['framebuffer_sha256'], d['blocking_render_multi'])"
timeout -k 10 600 python3 bench.py --multi abi --gpus 1 --steps 5 --warmup 1 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('abi 1 device', d['value'], d['ms_per_step'], d['framebuffer_sha256'], d['blocking_render_multi'])"
