#!/bin/bash
# GPU box: publication granularity of the inter-unit boundary (blocks of 6 rows), bench shape, 1 kb queries and reads
for pb in 0 5 10 20 40; do
  echo "PUB $pb (0 = the library's choice): "
  if [ $pb != 0 ]; then export SW_HIP_SCORE_PUB=$pb; fi
  timeout -k 10 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | grep -o "kernel_ms\": [0-9.]*" || exit 1
  timeout -k 10 200 python3 scratch/sweep.py c3 c1big c5 | grep -o "score_kernel_ms\": [0-9.]*" || exit 1
done
