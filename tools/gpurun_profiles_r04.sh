#!/bin/bash
# Round-4 profile collection on the GPU box (outputs under gpurun_out/prof4/, copied to profiles/ by hand).
# Headline workload: bench.py's default = C4, the 10^6-variable DOCP (K=200, nx=5000, nu=50), STAGED engine;
# the tree engine's configs (C2 banded system, C3 double-integrator QP) behind it.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof4; rm -rf $O; mkdir -p $O
# counter passes first (separate runs, --kernel-trace only; 40 stages: per-launch figures do not depend on the number of stages)
B="python3 bench.py --stages 40 --steps 1 --warmup 1 --no-cpu-baseline --no-ip"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O r04
cp $O/pmc_traffic.json $O/r04_pmc_traffic_c4.json; cp $O/r04_pmc_traffic_c4.json profiles/r04_pmc_traffic_c4.json
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- $B > /dev/null 2>&1
python tools/pmc_busy.py $O/pmc_mfma > $O/r04_pmc_mfma_busy.txt 2>&1
# the bench line (driver's command) and its kernel statistics
timeout 900 python bench.py --steps 20 --warmup 3 2>$O/r04_bench.err | grep '^{' | tail -1 > $O/r04_bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ip 2>/dev/null | grep '^{' | tail -1 > $O/r04_bench_under_rocprof.json
cp $(ls $O/kt/*/*kernel_stats.csv | tail -1) $O/r04_kernel_stats.csv
# C2 (tree engine): bench line, kernel statistics, launch by launch
timeout 300 python bench.py --workload c2 --steps 30 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/r04_bench_c2.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -- python3 bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-ip 2>/dev/null | grep '^{' | tail -1 > $O/r04_bench_c2_under_rocprof.json
cp $(ls $O/kt2/*/*kernel_stats.csv | tail -1) $O/r04_kernel_stats_c2.csv
timeout 300 bash tools/c2_trace.sh 160 > $O/r04_c2_timeline.txt 2>&1
timeout 300 python tools/stamps_top.py 2>&1 | grep -v amdgpu.ids > $O/r04_solve_top_stamps.txt
timeout 300 python tools/tree_levels.py 2>&1 | grep -v amdgpu.ids > $O/r04_c2_tree_levels.txt
timeout 300 python tools/block_time.py 2>&1 | grep -v amdgpu.ids > $O/r04_block_time.txt
# C3 (double-integrator QP, K = 2000): the device-resident Mehrotra loop per kernel
timeout 300 bash tools/ipprof.sh 2000 > $O/r04_ip_did_kstat.txt 2>&1
# STAGED engine, control-sized paths: stages beyond one CU of LDS and free initial states of 250 / 1000 components; the
# symmetric products of the solve (micro-benchmark of the tile forms, and the solve's kernels on ten stages of the headline)
timeout 300 python tools/bigstage_time.py 2>&1 | grep -v amdgpu.ids > $O/r04_bigstage_time.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o /tmp/symv_probe tools/symv_probe.hip && timeout 100 /tmp/symv_probe > $O/r04_symv.txt 2>&1
for t in tri rows; do
  if [ $t = rows ]; then export HQPKKT_NO_SYMV=1; fi
  rm -rf /tmp/p
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -- python3 tools/c4_bench.py 10 5000 50 2 > /tmp/p.log 2>&1
  echo "== the solve's product kernels, ten stages of the headline, two factor+solve: $t" >> $O/r04_symv.txt
  grep -i "gemv\|symv\|cols_finish\|\"Name\"" $(find /tmp/p -name '*kernel_stats.csv' | head -1) | cut -d, -f1-4,6,7 >> $O/r04_symv.txt
done
unset HQPKKT_NO_SYMV
# N > 1 path: bench.py starting its own two ranks on the one GPU (exchange staged through gloo: functional, not a measurement)
timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --stages 20 --steps 3 --warmup 1 --no-ip 2>/dev/null | grep '^{' | tail -1 > $O/r04_bench_2rank_shared.json
rm -rf $O/kt $O/kt2 $O/pmc_fetch $O/pmc_write $O/pmc_mfma gpurun_out/c2trace/kt gpurun_out/prof_ip/ip_results.db
ls -la $O
