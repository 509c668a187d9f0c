#!/bin/bash
# Round-1 profile collection on the GPU box (outputs under gpurun_out/prof/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof; rm -rf $O; mkdir -p $O
# the counter passes first: bench.py copies the dominant kernel's HBM bytes per launch from
# profiles/pmc_traffic.json into roofline.traffic
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --steps 20 --warmup 3 2>/dev/null | grep '^{' | tail -1 > $O/r01_bench.json
python bench.py --steps 20 --warmup 3 --host-vectors --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/r01_bench_hostptr.json
python bench.py --steps 20 --warmup 3 --mode RedSpBKP --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/r01_bench_redspbkp.json
python bench.py --n 400000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/r01_bench_n1e6.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/r01_bench_under_rocprof.json
cp $(ls $O/kt/*/*kernel_stats.csv | tail -1) $O/r01_kernel_stats.csv
# the device-resident Mehrotra loop on the Prg_DID structure (second half of the metric): kernel
# statistics and the kernel timeline of one iteration (launch gaps = host round trips)
for K in 2000 33333; do
  rocprofv3 --kernel-trace --stats -d $O/ip$K -o ip -- python3 tools/ip_profile.py $K RedSpBKP > $O/r01_ip_K${K}_run.txt 2>/dev/null
  python3 tools/kstat.py $O/ip$K/ip_results.db k_factor_diag_small 16 > $O/r01_ip_K${K}_kernels.txt 2>&1
  python3 tools/timeline.py $O/ip$K/ip_results.db > $O/r01_ip_K${K}_timeline.txt 2>&1
  rm -rf $O/ip$K
done
rm -rf $O/kt $O/pmc_fetch $O/pmc_write
ls -la $O
