cd $GRAFT_REPO_ROOT
timeout 60 tools/_build/blk_probe 2>&1 | head -3
timeout 600 python -m pytest tests/test_gpu_block.py -x -q 2>&1 | tail -5
timeout 300 python tools/block_time.py 2>&1 | grep -v amdgpu.ids | head -8
