cd $GRAFT_REPO_ROOT
timeout 60 tools/_build/blk_probe
