#!/bin/bash
# sweep of the queue plan of k_dgemm_tn_sk (HQPKKT_SPLIT_PLAN = % whole, % up to which halves, pieces of the rest, whole rounds only)
for plan in 45,80,8,1 90,90,8,0 85,85,8,0 80,80,8,0 95,95,8,0 90,90,4,0 90,90,16,0 70,90,8,0 60,85,8,0 64,64,8,1 62,93,9,1 96,96,8,1; do
  echo "== plan $plan"
  HQPKKT_SPLIT_PLAN=$plan python3 tools/dgemm_shapes.py 5000x5050x5000x0 5050x5050x5000x1x0 2>&1 | grep dgemm
done
