cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -DHQPKKT_STAMPS -o hqp_amd/libhqpkkt_stamps.so hqp_amd/csrc/hqpkkt.hip hqp_amd/csrc/analysis.cpp hqp_amd/csrc/staged_plan.cpp 2>&1 | grep -i "error" | head
HQPKKT_LIB=hqp_amd/libhqpkkt_stamps.so timeout 300 python tools/stamps_blk.py 160 2>&1 | head -64
