cd $GRAFT_REPO_ROOT
export SHARD_BACKEND=gloo MASTER_ADDR=127.0.0.1
SHARD_CASES='[["c4dense", 4, 2001, 400, "LQDOCP"]]' timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tests/shard_worker.py 2>&1 | grep -v "Gloo\|amdgpu.ids\|^$\|socket.cpp" | head -40
