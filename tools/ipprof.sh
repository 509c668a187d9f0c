#!/bin/bash
# usage: tools/ipprof.sh [K] [mode]  -> gpurun_out/prof_ip/{plain.log,ip_results.db,kstat.txt,timeline.txt}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
K=${1:-33333}; MODE=${2:-RedSpBKP}
mkdir -p gpurun_out/prof_ip
rm -f gpurun_out/prof_ip/*
python3 tools/ip_profile.py $K $MODE > gpurun_out/prof_ip/plain.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ip -o ip -- python3 tools/ip_profile.py $K $MODE > gpurun_out/prof_ip/rocprof.log 2>&1
python3 tools/kstat.py gpurun_out/prof_ip/ip_results.db k_factor_diag_small 30 > gpurun_out/prof_ip/kstat.txt 2>&1
python3 tools/timeline.py gpurun_out/prof_ip/ip_results.db > gpurun_out/prof_ip/timeline.txt 2>&1
cat gpurun_out/prof_ip/plain.log gpurun_out/prof_ip/kstat.txt
