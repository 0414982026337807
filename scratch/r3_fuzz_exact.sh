#!/bin/bash
# the fuzzers that reach the exact kernels and the walker, on the tree with the v_alignbit direction words
set -e
OUT=gpurun_out/r03_fuzz_exact.txt
: > $OUT
echo "tree: $(cat .git_head 2>/dev/null || echo unknown) (exact kernel: one line-aligned publication store per block)" >> $OUT
for seed in 801 802; do
  timeout -k 10 280 python3 scratch/fuzz.py $seed 30000 >> $OUT 2>&1
done
timeout -k 10 170 python3 scratch/fuzz_long.py 821 600 >> $OUT 2>&1
timeout -k 10 170 python3 scratch/fuzz_package.py 831 40 >> $OUT 2>&1
timeout -k 10 170 python3 scratch/fuzz_package.py 851 60 sets >> $OUT 2>&1
timeout -k 10 170 python3 scratch/fuzz_db.py 841 100 >> $OUT 2>&1
echo "all fuzzers done" >> $OUT
