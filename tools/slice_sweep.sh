#!/bin/bash
# column-slice shapes of a system sharded over 8 ranks (200 / 170 tiles): plain launch against cut pieces from a queue
echo "== default"; python3 tools/dgemm_shapes.py 5000x640x5000x0 4360x640x5000x0 640x640x5000x1 5000x1280x5000x0 2>&1 | grep dgemm
for plan in 0,0,2,0 0,0,3,0 0,0,4,0 0,0,6,0 0,0,8,0; do
  echo "== forced split, plan $plan"
  HQPKKT_FORCE_SPLIT=1 HQPKKT_SPLIT_PLAN=$plan python3 tools/dgemm_shapes.py 5000x640x5000x0 4360x640x5000x0 640x640x5000x1 5000x1280x5000x0 2>&1 | grep dgemm
done
