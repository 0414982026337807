#!/bin/bash
# GPU box: publication granularity of the inter-unit boundary (blocks of kScoreB rows), bench shape and C3
for pb in 5 10 20; do
  echo "PUB $pb: "
  SW_HIP_SCORE_PUB=$pb timeout -k 10 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | grep -o "kernel_ms\": [0-9.]*" || exit 1
  SW_HIP_SCORE_PUB=$pb timeout -k 10 200 python3 scratch/sweep.py c3 c1big | grep -o "score_kernel_ms\": [0-9.]*" || exit 1
done
