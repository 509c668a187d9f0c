#!/bin/bash
# Round-3 randomised campaign on the final kernels (outputs under gpurun_out/fuzz3/).
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz3; mkdir -p $O; F=$O/r03_fuzz_big.txt; : > $F
echo "## tools/fuzz_staged.py, 10000 cases in chunks of 400 (STAGED engine against the reference's Hqp_IpLQDOCP)" >> $F
for s0 in $(seq 0 400 9600); do python tools/fuzz_staged.py 400 $s0 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F; done
echo "## tools/fuzz.py 12000 (tree engine against the CPU oracle)" >> $F
python tools/fuzz.py 12000 2>/dev/null | grep -v amdgpu.ids | tail -8 >> $F
echo "## FUZZ_ORDERING=1 tools/fuzz.py 3000" >> $F
FUZZ_ORDERING=1 python tools/fuzz.py 3000 2>/dev/null | grep -v amdgpu.ids | tail -8 >> $F
echo "## tools/fuzz_ip.py, 6000 QPs in chunks of 400 (device loops against the reference's solvers)" >> $F
for s0 in $(seq 0 400 5600); do python tools/fuzz_ip.py 400 $s0 2>/dev/null | grep -v amdgpu.ids | tail -4 >> $F; done
echo "## tools/fuzz_bigstage.py, 1000 QPs in chunks of 100 (stages of 10 ... 300 controls against the tree engine)" >> $F
for s0 in $(seq 0 100 900); do python tools/fuzz_bigstage.py 100 $s0 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F; done
tail -5 $F
