cd $GRAFT_REPO_ROOT
for v in "-DFB_OWNSIMD=true" "-DFB_OWNSIMD=false"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function $v -o /tmp/libx.so hqp_amd/csrc/hqpkkt.hip hqp_amd/csrc/analysis.cpp hqp_amd/csrc/staged_plan.cpp 2>&1 | grep -i " error" | head -3
  echo "== variant [$v]"
  HQPKKT_LIB=/tmp/libx.so timeout 120 python tools/block_time1.py 80 128 160 192 2>&1 | tail -4
done
