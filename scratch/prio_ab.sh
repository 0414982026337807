#!/bin/bash
# GPU box: dynamic issue priority (p1) with hand-over blocks of 6 / 4 / 8 rows and rings of 6 / 9 / 4 blocks
cd "$(dirname "$0")/.."
run() {
  timeout -k 10 300 python3 scratch/sweep.py c5 c1big c3 | grep -o 'score_kernel_ms": [0-9.]*' || exit 1
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline | grep -o '"kernel_ms": [0-9.]*' || exit 1
}
export SW_HIP_SCORE_FEW_GROUPS=100000000
for v in p1d6 b4d9 b8d4; do
  export SW_HIP_LIB=$PWD/gpurun_variants/libsw_$v.so
  echo "-- $v"; run
done
