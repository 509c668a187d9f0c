#!/bin/bash
# Round-5 randomised campaign of the STAGED engine on the final code of the round (the solve's batched products with V, the
# fractional cut): outputs under gpurun_out/fuzz5/.  HQPKKT_SYMV_FROM=16 sends every stage width through the triangle form
# and therefore through the batched launches.
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz5; mkdir -p $O; F=$O/r05_fuzz_staged.txt; : > $F
# (chunks of 500: Meschach ends the process after its 100th caught error, and singular random QPs are errors it catches)
echo "## tools/fuzz_staged.py, cases 0 .. 3999 in chunks of 500 (against the reference's Hqp_IpLQDOCP)" >> $F
for s0 in $(seq 0 500 3500); do timeout 600 python tools/fuzz_staged.py 500 $s0 2>/dev/null | grep -v amdgpu.ids | tail -3 >> $F; done
echo "## HQPKKT_SYMV_FROM=16 tools/fuzz_staged.py, cases 4000 .. 5999 in chunks of 500" >> $F
for s0 in $(seq 4000 500 5500); do HQPKKT_SYMV_FROM=16 timeout 600 python tools/fuzz_staged.py 500 $s0 2>/dev/null | grep -v amdgpu.ids | tail -3 >> $F; done
echo "## tools/fuzz_bigstage.py 800 (stages of 10 ... 300 controls against the tree engine)" >> $F
timeout 1200 python tools/fuzz_bigstage.py 800 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F
echo "## HQPKKT_SYMV_FROM=16 tools/fuzz_bigstage.py 400 800" >> $F
HQPKKT_SYMV_FROM=16 timeout 900 python tools/fuzz_bigstage.py 400 800 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F
cat $F
