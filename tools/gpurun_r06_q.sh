#!/bin/bash
out=gpurun_out/r06_fused.txt
python3 tools/ip_profile.py 2000 RedSpBKP 2>&1 | grep "it/s" > $out
for B in 0 700; do HQPKKT_LIB=$PWD/tools/_build/libstamps_fds_$B.so timeout 120 python3 tools/stamps_small.py 2000 2>&1 | grep -v amdgpu | tail -2 >> $out; done
timeout 2000 python3 -m pytest tests/test_gpu_franke.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sweep.py -q -x 2>&1 | tail -3 >> $out
cat $out
