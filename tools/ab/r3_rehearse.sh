#!/bin/bash
# two and four ranks of bench.py on ONE GPU (gloo barrier, the product's gather pipeline, the abi child): the N > 1 code path end to end
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for n in 2 4; do
timeout -k 10 600 python3 bench.py --gpus $n --rehearse-one-gpu --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>gpurun_out/r3_rehearse_$n.err | tail -1 > gpurun_out/r3_rehearse_$n.json
python3 -c "
import sys,json; d=json.loads(open('gpurun_out/r3_rehearse_$n.json').read()); print('rehearse N=$n', d['value'], d['ms_per_step'], d['framebuffer_sha256'], d['n_gpus'], d['scaling']); print(json.dumps(d['multi_paths'])[:700])"
done
