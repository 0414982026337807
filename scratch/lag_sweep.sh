#!/bin/bash
# GPU box: number of active lanes S of the ticket order (SW_HIP_SCORE_LANES), all database shapes
for S in 1024 2048 4096 8192; do
  echo "S $S: "
  SW_HIP_SCORE_LANES=$S timeout -k 10 200 python3 scratch/sweep.py c1 c1big c2 c3 c5s gen:5000:100000:1000 gen:300:1000000:300 | grep -o "score_kernel_ms\": [0-9.]*" || exit 1
done
