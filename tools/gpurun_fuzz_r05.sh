#!/bin/bash
# Round-5 randomised campaign on the final code of the round (outputs under gpurun_out/fuzz5/).
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz5; mkdir -p $O; F=$O/r05_fuzz_tree.txt; : > $F
echo "## tools/fuzz.py 12000 (tree engine against the CPU oracle)" >> $F
timeout 1500 python tools/fuzz.py 12000 2>/dev/null | grep -v amdgpu.ids | tail -8 >> $F
echo "## FUZZ_SCALE=30 tools/fuzz.py 60 (banded systems up to n = 45 000, band 120: fronts of 100 .. 192 pivots)" >> $F
FUZZ_SCALE=30 timeout 1500 python tools/fuzz.py 60 2>/dev/null | grep -v amdgpu.ids | tail -8 >> $F
echo "## FUZZ_ORDERING=1 tools/fuzz.py 3000" >> $F
FUZZ_ORDERING=1 timeout 900 python tools/fuzz.py 3000 2>/dev/null | grep -v amdgpu.ids | tail -8 >> $F
echo "## tools/fuzz_ip.py, 12000 QPs in chunks of 400 (device loops against the reference's solvers)" >> $F
for s0 in $(seq 0 400 11600); do timeout 600 python tools/fuzz_ip.py 400 $s0 2>/dev/null | grep -v amdgpu.ids | tail -4 >> $F; done
echo "## FUZZ_HOT=1 tools/fuzz_ip.py 800 (hot starts)" >> $F
FUZZ_HOT=1 timeout 600 python tools/fuzz_ip.py 800 2>/dev/null | grep -v amdgpu.ids | tail -4 >> $F
tail -60 $F
