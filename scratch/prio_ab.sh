#!/bin/bash
# GPU box: issue priorities in the 4-wave workgroups of small passes too?  (three alternating rounds: the spread over runs is large)
cd "$(dirname "$0")/.."
for r in 1 2 3; do
for v in "" w4p; do
  if [ -n "$v" ]; then export SW_HIP_LIB=$PWD/gpurun_variants/libsw_$v.so; else unset SW_HIP_LIB; fi
  echo "-- '$v'"
  timeout -k 10 100 python3 scratch/lanes15.py 1 | tail -3 | tr '\n' ' '; echo
  timeout -k 10 100 python3 scratch/lanes15.py 15 | tail -3 | tr '\n' ' '; echo
  timeout -k 10 100 python3 scratch/bench_rust.py | grep -o "= [0-9.]* GCUPS" | tr '\n' ' '; echo
  timeout -k 10 100 python3 scratch/sweep.py c1 | grep -o '"search_ms": [0-9.]*'
done
done
