#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_solve_top.py -x -q 2>&1 | tail -12
timeout 250 python tools/scratch/did.py 2>&1 | grep -v amdgpu.ids | tail -4
HQPKKT_NO_TREE_FACTOR=1 timeout 250 python tools/scratch/did.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_reference_host.py -x -q 2>&1 | tail -5
