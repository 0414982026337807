#!/bin/bash
# GPU box: pipeline stages of a SIMD's four waves adjacent (perm) against interleaved (product)
cd "$(dirname "$0")/.."
run() {
  timeout -k 10 300 python3 scratch/sweep.py c5 c1big c3 | grep -o 'score_kernel_ms": [0-9.]*\|"top": [0-9]*' | tr '\n' ' '; echo
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline | grep -o '"kernel_ms": [0-9.]*' || exit 1
}
echo "-- product"; run
export SW_HIP_LIB=$PWD/gpurun_variants/libsw_perm.so
echo "-- perm"; run
