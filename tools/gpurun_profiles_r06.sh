#!/bin/bash
# Round-6 profile collection on the GPU box (outputs under gpurun_out/prof6/, copied to profiles/ by hand).
# Headline workload: bench.py's default = C4, the 10^6-variable DOCP (K=200, nx=5000, nu=50), STAGED engine;
# the tree engine's configs (C2 banded system, C3 double-integrator QP) and the stand-ins of configs[4] behind it.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof6; rm -rf $O; mkdir -p $O
# counter passes first (separate runs, --kernel-trace only; 40 stages: per-launch figures do not depend on the number of stages)
B="python3 bench.py --stages 40 --steps 1 --warmup 1 --no-cpu-baseline --no-ip"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O r06
cp $O/pmc_traffic.json $O/r06_pmc_traffic_c4.json; cp $O/r06_pmc_traffic_c4.json profiles/r06_pmc_traffic_c4.json
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- $B > /dev/null 2>&1
python tools/pmc_busy.py $O/pmc_mfma > $O/r06_pmc_mfma_busy.txt 2>&1
# the bench line (driver's command) and its kernel statistics
timeout 900 python bench.py --steps 20 --warmup 3 2>$O/r06_bench.err | grep '^{' | tail -1 > $O/r06_bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ip 2>/dev/null | grep '^{' | tail -1 > $O/r06_bench_under_rocprof.json
cp $(ls $O/kt/*/*kernel_stats.csv | tail -1) $O/r06_kernel_stats.csv
# C2 (tree engine): bench line, kernel statistics, launch by launch, counters, stamps inside the pivot-block and the panel kernels
timeout 300 python bench.py --workload c2 --steps 30 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/r06_bench_c2.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -- python3 bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-ip 2>/dev/null | grep '^{' | tail -1 > $O/r06_bench_c2_under_rocprof.json
cp $(ls $O/kt2/*/*kernel_stats.csv | tail -1) $O/r06_kernel_stats_c2.csv
timeout 300 bash tools/c2_trace.sh 160 > $O/r06_c2_timeline.txt 2>&1
timeout 600 bash tools/pmc_tree.sh > $O/r06_pmc_tree.txt 2>&1
if [ -f hqp_amd/libhqpkkt_stamps.so ]; then
  HQPKKT_LIB=$PWD/hqp_amd/libhqpkkt_stamps.so timeout 120 python3 tools/stamps_fb.py 160 qd 2>&1 | grep -v amdgpu > $O/r06_fb_panel_stamps.txt
  HQPKKT_LIB=$PWD/hqp_amd/libhqpkkt_stamps.so timeout 120 python3 tools/stamps_ps.py 2>&1 | grep -v amdgpu > $O/r06_ps_stamps.txt
fi
timeout 300 python3 tools/block_time.py 2>&1 | grep -v amdgpu > $O/r06_block_time.txt
# C3 (double-integrator QP, K = 2000): the device-resident Mehrotra loop per kernel
timeout 300 bash tools/ipprof.sh 2000 > $O/r06_ip_did_kstat.txt 2>&1
cp gpurun_out/prof_ip/timeline.txt $O/r06_ip_did_timeline.txt 2>/dev/null
timeout 60 tools/post_probe > $O/r06_post_probe.txt 2>&1
# ... the same loop launch by launch (no segment graphs), the bench line's IP section, and the reference's loop on the plugin
# through the shim (calls with host vectors): it/s with and without the call graphs / kernel copies, one iteration launch by launch
{ echo "== hqpkkt_mehrotra, Prg_DID K = 2000 (tools/ip_profile.py)"; python3 tools/ip_profile.py 2000 RedSpBKP 2>&1 | grep -v amdgpu
  echo "== HQPKKT_NO_IP_SEGMENTS=1"; HQPKKT_NO_IP_SEGMENTS=1 python3 tools/ip_profile.py 2000 RedSpBKP 2>&1 | grep -v amdgpu; } > $O/r06_ip_gaps.txt
{ echo "== the reference's Hqp_IpsMehrotra on RedSpBKPHip through the shim (tools/shim_profile.py 2000 5)"; python3 tools/shim_profile.py 2000 5 2>&1 | grep -v amdgpu
  echo "== HQPKKT_NO_HOST_GRAPHS=1"; HQPKKT_NO_HOST_GRAPHS=1 python3 tools/shim_profile.py 2000 5 2>&1 | grep -v amdgpu
  echo "== HQPKKT_NO_HOST_GRAPHS=1 HQPKKT_NO_HOST_KERNEL_COPIES=1 (the chain of round 5 but for the posted status words)"; HQPKKT_NO_HOST_GRAPHS=1 HQPKKT_NO_HOST_KERNEL_COPIES=1 python3 tools/shim_profile.py 2000 5 2>&1 | grep -v amdgpu; } > $O/r06_shim.txt
mkdir -p $O/shim
for v in graphs chain; do
  rm -rf $O/shim/*
  if [ $v = chain ]; then export HQPKKT_NO_HOST_GRAPHS=1 HQPKKT_NO_HOST_KERNEL_COPIES=1; fi
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/shim -o shim -- python3 tools/shim_profile.py 2000 2 > /dev/null 2>&1
  unset HQPKKT_NO_HOST_GRAPHS HQPKKT_NO_HOST_KERNEL_COPIES
  { echo "== one iteration of the reference's loop on the plugin, $v"; python3 tools/shim_timeline.py $O/shim/shim_results.db 40; } >> $O/r06_shim_timeline.txt 2>&1
done
rm -rf $O/shim
# stamps inside the whole-tree launch of the small-front kernel (builds with -DHQPKKT_STAMPS -DFSTAMP_GRID=1023 -DFSTAMP_BLOCK=..: tools/_build)
for B in 0 700 1000 1022; do
  if [ -f tools/_build/libstamps_fds_$B.so ]; then
    echo "== block $B of 1023" >> $O/r06_fds_tree_stamps.txt
    HQPKKT_LIB=$PWD/tools/_build/libstamps_fds_$B.so timeout 120 python3 tools/stamps_small.py 2000 2>&1 | grep -v amdgpu >> $O/r06_fds_tree_stamps.txt
  fi
done
# the cut form of the fp64 product: equal shares against the work table, and the table's stamps
{ for t in 0 1 0 1; do echo "== HQPKKT_SK_TABLE=$t"; HQPKKT_SK_TABLE=$t python3 tools/dgemm_stamps.py 5000x5050x5000x0 5050x5050x5000x1 3000x3050x3000x0 2>&1 | grep -v amdgpu; done; } > $O/r06_sk_table.txt
HQPKKT_DGEMM_STAMPS=1 python3 tools/dgemm_stamps.py 5000x5050x5000x0 5050x5050x5000x1 2>&1 | grep -v amdgpu > $O/r06_sk_stamps.txt
HQPKKT_SK_TABLE=0 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ip 2>/dev/null | grep '^{' | tail -1 > $O/r06_bench_equal_shares.json
# configs[4] stand-ins (mesh 300 x 300, 1000 x 1000, band of 21 with 1000 far couplings at 10^5 variables, the mesh with 1 % far couplings)
timeout 600 python tools/mesh_bench.py 2>/dev/null | grep '^{' | tail -1 > $O/r06_mesh_bench.json
# SURVEY C5's row density (10 ... 100 entries per row) at 1e5 variables
timeout 600 python3 tools/c5_cute.py 100000 2>/dev/null | grep '^{' > $O/r06_c5_cute.jsonl
# configs[4] at full size on the irregular generator: the SQP loop at 10^6 variables with 1 % far couplings
timeout 900 python tools/c5_irregular.py 2>/dev/null | grep '^{' > $O/r06_c5_irregular.jsonl
# mid-size stages
for nx in 1000 2000 3000; do timeout 300 python tools/c4_bench.py 200 $nx 50 3 2>/dev/null | grep '^{' | tail -1 >> $O/r06_c4_sizes.jsonl; done
# N > 1 path: bench.py starting its own two ranks on the one GPU (exchange staged through gloo: functional, not a measurement)
timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --stages 20 --steps 3 --warmup 1 --no-ip 2>/dev/null | grep '^{' | tail -1 > $O/r06_bench_2rank_shared.json
# one rank of P on the one GPU with a transport that moves nothing: the pieces of the sharded stage (tools/shard_model.py)
timeout 600 bash tools/slice_products.sh > $O/r06_slice_products.txt 2>&1
echo "## the model (tools/shard_model.py)" >> $O/r06_slice_products.txt
python3 tools/shard_model.py $O/r06_slice_products.txt >> $O/r06_slice_products.txt 2>&1
rm -rf $O/kt $O/kt2 $O/pmc_fetch $O/pmc_write $O/pmc_mfma gpurun_out/c2trace/kt gpurun_out/prof_ip/ip_results.db
ls -la $O
