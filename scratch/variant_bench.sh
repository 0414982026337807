#!/bin/bash
# GPU box: bench kernel time for each library variant under scratch/variants
for f in scratch/variants/*.so; do
  echo -n "$f: "
  SW_HIP_LIB=$PWD/$f timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | grep -o "kernel_ms\": [0-9.]*"
done
echo -n "default: "; timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | grep -o "kernel_ms\": [0-9.]*"
