#!/bin/bash
# Round-5 randomised campaign of the STAGED engine on the final code of the round (the solve's batched products with V, the
# fractional cut): outputs under gpurun_out/fuzz5/.  HQPKKT_SYMV_FROM=16 sends every stage width through the triangle form
# and therefore through the batched launches.
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz5; mkdir -p $O; F=$O/r05_fuzz_staged.txt; : > $F
echo "## tools/fuzz_staged.py 4000 (against the reference's Hqp_IpLQDOCP)" >> $F
timeout 1200 python tools/fuzz_staged.py 4000 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F
echo "## HQPKKT_SYMV_FROM=16 tools/fuzz_staged.py 2000 4000" >> $F
HQPKKT_SYMV_FROM=16 timeout 900 python tools/fuzz_staged.py 2000 4000 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F
echo "## tools/fuzz_bigstage.py 800 (stages of 10 ... 300 controls against the tree engine)" >> $F
timeout 1200 python tools/fuzz_bigstage.py 800 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F
echo "## HQPKKT_SYMV_FROM=16 tools/fuzz_bigstage.py 400 800" >> $F
HQPKKT_SYMV_FROM=16 timeout 900 python tools/fuzz_bigstage.py 400 800 2>/dev/null | grep -v amdgpu.ids | tail -6 >> $F
cat $F
