#!/bin/bash
# Round-3 profile collection on the GPU box (outputs under gpurun_out/prof3/, copied to profiles/ by hand).
# Workload: bench.py's default = C4, the 10^6-variable DOCP (K=200, nx=5000, nu=50), STAGED engine.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof3; rm -rf $O; mkdir -p $O
# counter passes first (separate runs, --kernel-trace only; 40 stages: per-launch figures do not depend on the number of stages)
B="python3 bench.py --stages 40 --steps 1 --warmup 1 --no-cpu-baseline --no-ip"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O r03
cp $O/pmc_traffic.json $O/r03_pmc_traffic_c4.json; cp $O/r03_pmc_traffic_c4.json profiles/r03_pmc_traffic_c4.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- $B > /dev/null 2>&1
python tools/pmc_busy.py $O/pmc_mfma > $O/r03_pmc_mfma_busy.txt 2>&1
# the bench line (driver's command) and its kernel statistics
python bench.py --steps 20 --warmup 3 2>$O/r03_bench.err | grep '^{' | tail -1 > $O/r03_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ip 2>/dev/null | grep '^{' | tail -1 > $O/r03_bench_under_rocprof.json
cp $(ls $O/kt/*/*kernel_stats.csv | tail -1) $O/r03_kernel_stats.csv
# other sizes of the same structure (K = 200), a stage with 200 controls at full width, round 1's headline workload
for nx in 1000 2000 3000; do python tools/c4_bench.py 200 $nx 50 3 --profile 2>/dev/null | grep '^{' | tail -1 >> $O/r03_c4_sizes.jsonl; done
# (the same without the per-launch events of --profile: the rates DESIGN.md quotes)
for nx in 400 700 1000 1500 2000 2500 3000; do python tools/c4_bench.py 200 $nx 50 3 2>/dev/null | grep '^{' | tail -1 >> $O/r03_c4_sizes_plain.jsonl; done
./tools/mid_trace.sh 1000 > $O/r03_stage_timeline_nx1000.txt 2>&1
python tools/c4_bench.py 200 5000 100 2 2>/dev/null | grep '^{' | tail -1 > $O/r03_c4_nu100.json
python tools/c4_bench.py 200 5000 200 2 2>/dev/null | grep '^{' | tail -1 > $O/r03_c4_nu200.json
python tools/bigstage_time.py 2>&1 | grep -v amdgpu.ids > $O/r03_bigstage_time.txt
python bench.py --workload c2 --steps 20 --warmup 3 2>/dev/null | grep '^{' | tail -1 > $O/r03_bench_c2.json
# the fp64 product on its own: rates by shape (three staging variants), time stamps per workgroup
python tools/dgemm_ab.py 2>&1 | grep -v amdgpu.ids > $O/r03_dgemm_sizes.txt
HQPKKT_DGEMM_STAMPS=1 python tools/dgemm_stamps.py 5000x5050x5000x0 5050x5050x5000x1 4096x4096x4096x0 2>&1 | grep -v amdgpu.ids > $O/r03_dgemm_stamps.txt
# N > 1 path: bench.py starting its own two ranks on the one GPU (exchange staged through gloo: functional, not a measurement)
python bench.py --gpus 2 --backend gloo --share-gpu --stages 20 --steps 3 --warmup 1 --no-ip 2>/dev/null | grep '^{' | tail -1 > $O/r03_bench_2rank_shared.json
# single-GPU pieces of the multi-GPU model: the column-slice products of 2 / 4 / 8 ranks
python tools/dgemm_shapes.py 5000x2560x5000x0 5000x2560x5000x1 5000x1280x5000x0 4360x1280x5000x1 3720x1280x5000x1 5000x640x5000x0 5000x640x5000x1 2440x640x5000x1 5000x5000x50x1x1 5000x50x5000x0 50x5050x5000x0 2>&1 | grep dgemm > $O/r03_slice_products.txt
# the randomised sweep of the STAGED engine against the reference's Hqp_IpLQDOCP, in chunks
for s0 in $(seq 0 400 9600); do python tools/fuzz_staged.py 400 $s0 2>/dev/null | grep -v amdgpu.ids | tail -12; done > $O/r03_fuzz_staged.txt
rm -rf $O/kt $O/pmc_fetch $O/pmc_write $O/pmc_mfma
ls -la $O
