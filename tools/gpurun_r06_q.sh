#!/bin/bash
out=gpurun_out/r06_sing.txt
python3 - > $out 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import fuzz_ip
for c in (187, 6258, 2536, 8650, 193, 7260, 7511):
    print(fuzz_ip.check(c))
PY
timeout 2000 python3 -m pytest tests/test_reference_host.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_franke.py -q -x 2>&1 | tail -5 >> $out
grep -v amdgpu $out | tail -16 | cut -c1-200
