#!/bin/bash
cd "$GRAFT_REPO_ROOT"; timeout -k 10 120 python3 tools/ab/prog_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
