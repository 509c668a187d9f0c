#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/shim
out=gpurun_out/r06_franke_shim.txt
echo "== franke tests" > $out
timeout 1200 python3 -m pytest tests/test_gpu_franke.py tests/test_gpu_fuzz.py -q -x -rx 2>&1 | tail -12 >> $out
echo "== bench ip section" >> $out
python3 - >> $out 2>&1 <<'PY'
import json, bench
print(json.dumps(bench.ip_iterations(2000), indent=1))
PY
echo "== shim" >> $out
python3 tools/shim_profile.py 2000 4 >> $out 2>&1
rm -rf gpurun_out/shim/*
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/shim -o shim -- python3 tools/shim_profile.py 2000 2 > gpurun_out/shim/rocprof.log 2>&1
python3 tools/shim_timeline.py gpurun_out/shim/shim_results.db >> $out 2>&1
grep -v amdgpu $out | tail -120
