#!/bin/bash
# Round-6 randomised campaign on the final code of the round (outputs under gpurun_out/fuzz6/).
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz6; mkdir -p $O; F=$O/r06_fuzz_final.txt; : > $F
echo "## tools/fuzz.py 6000 (tree engine against the CPU oracle)" >> $F
timeout 1200 python tools/fuzz.py 6000 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F
echo "## FUZZ_SCALE=30 tools/fuzz.py 40 (banded systems up to n = 45 000, band 120: fronts of 100 .. 192 pivots)" >> $F
FUZZ_SCALE=30 timeout 1200 python tools/fuzz.py 40 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F
echo "## FUZZ_ORDERING=2 tools/fuzz.py 1500" >> $F
FUZZ_ORDERING=2 timeout 900 python tools/fuzz.py 1500 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F
echo "## tools/fuzz_ip.py, 8000 QPs in chunks of 400 (device loops - whole segments of an iteration as graphs - against the reference's solvers)" >> $F
for s0 in $(seq 0 400 7600); do timeout 600 python tools/fuzz_ip.py 400 $s0 2>/dev/null | grep -v amdgpu.ids | tail -4 >> $F; done
echo "## FUZZ_HOT=1 tools/fuzz_ip.py, 2000 hot-started QPs in chunks of 400" >> $F
for s0 in $(seq 0 400 1600); do FUZZ_HOT=1 timeout 600 python tools/fuzz_ip.py 400 $s0 2>/dev/null | grep -v amdgpu.ids | tail -4 >> $F; done
echo "## tools/fuzz_staged.py 2000 (STAGED engine, the table form of the cut products included where it applies)" >> $F
timeout 1200 python tools/fuzz_staged.py 2000 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F
echo "## tools/fuzz_sqp.py 60 (the reference's SQP loop on the shim: calls with host vectors as graphs)" >> $F
timeout 900 python tools/fuzz_sqp.py 60 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F
tail -80 $F
