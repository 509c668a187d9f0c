#!/bin/bash
out=gpurun_out/r06_segtest.txt
timeout 1500 python3 -m pytest tests/test_gpu_franke.py -q -k "segment or host_vectors" --tb=short 2>&1 | tail -40 > $out
grep -v amdgpu $out | cut -c1-400
