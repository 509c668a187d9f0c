mkdir -p gpurun_out
{
echo "# large campaign, final code of round 2"
echo "## tools/fuzz.py 24000 (tree engine against the CPU oracle)"
python tools/fuzz.py 24000 0 2>&1 | grep -v amdgpu | tail -1
echo "## FUZZ_ORDERING=1 tools/fuzz.py 6000"
FUZZ_ORDERING=1 python tools/fuzz.py 6000 0 2>&1 | grep -v amdgpu | tail -1
echo "## tools/fuzz_staged.py, 10000 cases in chunks of 400 (STAGED engine against the reference's Hqp_IpLQDOCP)"
for s in $(seq 0 400 9600); do python tools/fuzz_staged.py 400 $s 2>/dev/null | grep -E "BAD|fuzz_staged:" ; done
echo "## tools/fuzz_ip.py, 12000 QPs in chunks of 400 (device loops against the reference's solvers)"
for s in $(seq 0 400 11600); do python tools/fuzz_ip.py 400 $s 2>&1 | grep -E "MISMATCH|cases from"; done
} > gpurun_out/r02_fuzz_big.txt 2>&1
grep -c MISMATCH gpurun_out/r02_fuzz_big.txt; grep -E "^##|bad,|BAD': [1-9]" gpurun_out/r02_fuzz_big.txt | head -80 | tail -40
