#!/bin/bash
# Round 3: the fuzzers on the final tree, totals kept in gpurun_out/r03_fuzz.txt (copied to profiles/).
# Each run under its own time limit; stops at the first failure.
set -e
OUT=gpurun_out/r03_fuzz.txt
: > $OUT
git_head=$(cat .git_head 2>/dev/null || echo unknown)
echo "tree: $git_head" >> $OUT
for seed in 601 602 603; do
  timeout -k 10 280 python3 scratch/fuzz.py $seed 30000 >> $OUT 2>&1
done
echo "--- SW_HIP_LANES=1 (one-pair-per-lane kernel forced for every eligible pair)" >> $OUT
for seed in 611 612; do
  SW_HIP_LANES=1 timeout -k 10 280 python3 scratch/fuzz.py $seed 30000 >> $OUT 2>&1
done
timeout -k 10 170 python3 scratch/fuzz_long.py 621 600 >> $OUT 2>&1
# (forced onto single long pairs the lane kernel runs one wave per pair: a few dozen are enough)
SW_HIP_LANES=1 timeout -k 10 170 python3 scratch/fuzz_long.py 622 24 >> $OUT 2>&1
timeout -k 10 170 python3 scratch/fuzz_package.py 631 40 >> $OUT 2>&1
# (batches in which every pair has a target of its own: the context lanes of the fast path)
timeout -k 10 170 python3 scratch/fuzz_package.py 651 60 sets >> $OUT 2>&1
timeout -k 10 170 python3 scratch/fuzz_db.py 641 100 >> $OUT 2>&1
echo "all fuzzers done" >> $OUT
