cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_block.py -x -q -s 2>&1 | tail -30
timeout 300 python tools/block_time.py 2>&1 | tail -20
