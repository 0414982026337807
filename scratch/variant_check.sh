#!/bin/bash
# GPU box: database parity tests, then the bench kernel time, for each library variant under scratch/variants
for f in scratch/variants/*.so; do
  echo "== $f"
  SW_HIP_LIB=$PWD/$f timeout -k 10 200 python -u -m pytest tests/test_gpu_db_parity.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
  SW_HIP_LIB=$PWD/$f timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | grep -o "kernel_ms\": [0-9.]*" || exit 1
done
echo -n "default: "; timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | grep -o "kernel_ms\": [0-9.]*"
