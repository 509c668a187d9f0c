#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/shim
out=gpurun_out/r06_franke_shim.txt
echo "== shim" > $out
python3 tools/shim_profile.py 2000 5 >> $out 2>&1
echo "== shim, HQPKKT_NO_HOST_GRAPHS=1" >> $out
HQPKKT_NO_HOST_GRAPHS=1 python3 tools/shim_profile.py 2000 5 >> $out 2>&1
echo "== tests (host-vector paths)" >> $out
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_gpu_solve_top.py -q -x 2>&1 | tail -5 >> $out
rm -rf gpurun_out/shim/*
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/shim -o shim -- python3 tools/shim_profile.py 2000 2 > gpurun_out/shim/rocprof.log 2>&1
python3 tools/shim_timeline.py gpurun_out/shim/shim_results.db 40 >> $out 2>&1
grep -v amdgpu $out | tail -70
