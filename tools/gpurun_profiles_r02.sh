#!/bin/bash
# Round-2 profile collection on the GPU box (outputs under gpurun_out/prof2/, copied to profiles/).
# Workload: bench.py's default = C4, the 10^6-variable DOCP (K=200, nx=5000, nu=50), STAGED engine.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof2; rm -rf $O; mkdir -p $O
# counter passes first (separate runs, --kernel-trace only): bench.py copies k_dgemm_tn's HBM bytes per
# launch from profiles/r02_pmc_traffic_c4.json into roofline.traffic.  40 stages: per-launch figures do
# not depend on the number of stages
B="python3 bench.py --stages 40 --steps 1 --warmup 1 --no-cpu-baseline --no-ip"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O r02
cp $O/pmc_traffic.json profiles/r02_pmc_traffic_c4.json
# matrix-pipe and vector-pipe busy cycles of the same kernels
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- $B > /dev/null 2>&1
python tools/pmc_busy.py $O/pmc_mfma > $O/r02_pmc_mfma_busy.txt 2>&1
# the bench line (driver's command) and its kernel statistics
python bench.py --steps 20 --warmup 3 2>$O/r02_bench.err | grep '^{' | tail -1 > $O/r02_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ip 2>/dev/null | grep '^{' | tail -1 > $O/r02_bench_under_rocprof.json
cp $(ls $O/kt/*/*kernel_stats.csv | tail -1) $O/r02_kernel_stats.csv
# other sizes of the same structure (K = 200) and round 1's headline workload
for nx in 1000 2000 3000; do python tools/c4_bench.py 200 $nx 50 3 --profile 2>/dev/null | grep '^{' | tail -1 >> $O/r02_c4_sizes.jsonl; done
python bench.py --workload c2 --steps 20 --warmup 3 2>/dev/null | grep '^{' | tail -1 > $O/r02_bench_c2.json
python tools/staged_probe.py gemm 2>/dev/null > $O/r02_dgemm_sizes.txt
python tools/update_time.py 2>/dev/null | grep '^{' | tail -1 > $O/r02_update_time.json
# configs[4] stand-in: the sparse NLP through the reference's SQP host (reference's RedSpBKP up to 150 x 150 cells)
python tools/c5_bench.py 150 60 100 150 200 300 500 700 1000 2>/dev/null | grep '^{' > $O/r02_c5_grid_sqp.jsonl
# the randomised sweep of the STAGED engine against the reference's Hqp_IpLQDOCP, in chunks (the reference's
# Meschach error counter ends the process after 100 caught errors)
for s0 in 0 400 800 1200 1600 2000 2400 2800; do python tools/fuzz_staged.py 400 $s0 2>/dev/null | grep -v amdgpu.ids | tail -12; done > $O/r02_fuzz_staged.txt
rm -rf $O/kt $O/pmc_fetch $O/pmc_write $O/pmc_mfma
ls -la $O
