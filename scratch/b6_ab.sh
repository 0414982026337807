#!/bin/bash
# score kernel with 6-row hand-over blocks (gpurun_variants/libsw_b6.so) against the product's 12
for lib in "" $PWD/gpurun_variants/libsw_b6.so; do
  echo "== ${lib:-product library (B = 12)}"
  SW_HIP_LIB=$lib timeout -k 10 100 python3 -u scratch/lanes15.py 1 | tail -1
  SW_HIP_LIB=$lib timeout -k 10 100 python3 -u scratch/lanes15.py 15 | tail -1
  SW_HIP_LIB=$lib timeout -k 10 170 python3 scratch/fuzz_package.py 331 20 | tail -1
  SW_HIP_LIB=$lib timeout -k 10 200 python3 -u scratch/sweep.py c1 c5 b3 2>&1 | cut -c1-140
  SW_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-220
done
