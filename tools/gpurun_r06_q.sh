#!/bin/bash
out=gpurun_out/r06_fused.txt
echo "== fused" > $out
python3 tools/ip_profile.py 2000 RedSpBKP 2>&1 | grep "it/s" >> $out
echo "== HQPKKT_NO_FUSED_VECTORS=1" >> $out
HQPKKT_NO_FUSED_VECTORS=1 python3 tools/ip_profile.py 2000 RedSpBKP 2>&1 | grep "it/s" >> $out
echo "== shim" >> $out
python3 tools/shim_profile.py 2000 4 2>&1 | grep "it/s" >> $out
timeout 2000 python3 -m pytest tests/test_gpu_franke.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_shard.py tests/test_gpu_sweep.py -q -x 2>&1 | tail -5 >> $out
cat $out
