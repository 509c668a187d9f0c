#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r06_ip_gaps.txt
echo "== segments" > $out
python3 tools/ip_profile.py 2000 RedSpBKP >> $out 2>&1
echo "== HQPKKT_NO_IP_SEGMENTS=1" >> $out
HQPKKT_NO_IP_SEGMENTS=1 python3 tools/ip_profile.py 2000 RedSpBKP >> $out 2>&1
echo "== tests" >> $out
timeout 2000 python3 -m pytest tests/test_gpu_franke.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_solve_top.py tests/test_gpu_sweep.py -q -x 2>&1 | tail -8 >> $out
bash tools/ipprof.sh 2000 > /dev/null 2>&1
cat gpurun_out/prof_ip/timeline.txt >> $out
grep -v amdgpu $out | tail -60
