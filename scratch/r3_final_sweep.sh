#!/bin/bash
# GPU box: the sweep, the literal drop-in and the capacity protocol on the final tree (profiles/r03_*)
cd "$(dirname "$0")/.."
timeout -k 10 600 python3 scratch/sweep.py c1 c1big c3 c5 c4 c4m b1 b2 b5 b3 w2 w5 w5m > gpurun_out/r03_sweep.jsonl 2> gpurun_out/r03_sweep.err || exit 1
timeout -k 10 200 python3 scratch/sweep.py w5e >> gpurun_out/r03_sweep.jsonl 2>> gpurun_out/r03_sweep.err || exit 1
timeout -k 10 100 python3 scratch/bench_rust.py > gpurun_out/r03_capacity.txt 2>&1 || exit 1
timeout -k 10 400 python3 scratch/dropin_bench.py > gpurun_out/r03_dropin.txt 2> gpurun_out/r03_dropin.err || exit 1
