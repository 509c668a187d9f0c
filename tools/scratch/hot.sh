#!/bin/bash
cd $GRAFT_REPO_ROOT
T='tests/test_reference_host.py::test_franke_hot_start_follows_the_reference'
timeout 300 python -m pytest "$T" -x -q 2>&1 | grep -E "assert|passed|failed" | head -5
echo "== two reads"; HQPKKT_FRANKE_TWO_READS=1 timeout 300 python -m pytest "$T" -x -q 2>&1 | grep -E "assert|passed|failed" | head -5
echo "== no tree"; HQPKKT_NO_TREE_SWEEPS=1 timeout 300 python -m pytest "$T" -x -q 2>&1 | grep -E "assert|passed|failed" | head -5
echo "== no tree factor"; HQPKKT_NO_TREE_FACTOR=1 timeout 300 python -m pytest "$T" -x -q 2>&1 | grep -E "assert|passed|failed" | head -5
