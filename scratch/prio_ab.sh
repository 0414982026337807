#!/bin/bash
# GPU box: variants of the priority map / back-off of the score kernel (scratch/score_variant.sh)
cd "$(dirname "$0")/.."
run() {
  timeout -k 10 300 python3 scratch/sweep.py c5 c1big c3 | grep -o 'score_kernel_ms": [0-9.]*' | tr '\n' ' ' || exit 1
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline | grep -o '"kernel_ms": [0-9.]*' || exit 1
}
for v in "" m1 m2 m3 m4 s8 ""; do
  if [ -n "$v" ]; then export SW_HIP_LIB=$PWD/gpurun_variants/libsw_$v.so; else unset SW_HIP_LIB; fi
  echo "-- '$v'"; run
done
export SW_HIP_LIB=$PWD/gpurun_variants/libsw_tl.so
timeout -k 10 200 python3 scratch/score_timeline.py 150 10000000 150 > gpurun_out/score_tl_reads.txt 2>&1 || exit 1
timeout -k 10 200 python3 scratch/score_timeline.py 10000 100000 1000 > gpurun_out/score_tl_c2.txt 2>&1 || exit 1
