#!/bin/bash
cd "$(dirname "$0")/.."
for lib in "" reorder; do
  if [ -n "$lib" ]; then export SW_HIP_LIB=$PWD/gpurun_variants/libsw_$lib.so; else unset SW_HIP_LIB; fi
  for rpl in 1 2 4; do
    echo "== lib '$lib' rpl $rpl"; SW_HIP_EXACT_RPL=$rpl SW_HIP_PACKAGE=0 timeout -k 10 100 python3 scratch/step_lag.py || exit 1
  done
done
