#!/bin/bash
# Round-4 randomised campaign of the STAGED engine's control-sized paths on the final code of the round: stages of many
# controls / carried rows (blocked elimination of K), free initial states of hundreds of components (blocked inverse of
# [V_0 B_0'; B_0 0], new in round 4), and - with HQPKKT_SYMV_FROM=16 - the triangle form of the products with V on
# every stage width.  Against the tree engine on the same QPs (outputs under gpurun_out/fuzz4/).
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz4; mkdir -p $O; F=$O/r04_fuzz_bigstage.txt; : > $F
echo "## FUZZ_X0=1 tools/fuzz_bigstage.py 150 (free initial states of 140 .. 1600 components)" >> $F
FUZZ_X0=1 timeout 1500 python tools/fuzz_bigstage.py 150 2>/dev/null | grep -v amdgpu.ids | tail -12 >> $F
echo "## HQPKKT_SYMV_FROM=16 tools/fuzz_bigstage.py 200 (stages of 10 .. 300 controls; the products with V read one triangle)" >> $F
HQPKKT_SYMV_FROM=16 timeout 1500 python tools/fuzz_bigstage.py 200 2>/dev/null | grep -v amdgpu.ids | tail -12 >> $F
echo "## HQPKKT_SYMV_FROM=16 tools/fuzz_staged.py 2000 (small stages, reference / oracle comparisons as in round 3)" >> $F
HQPKKT_SYMV_FROM=16 timeout 1500 python tools/fuzz_staged.py 2000 2>/dev/null | grep -v amdgpu.ids | tail -8 >> $F
tail -40 $F
