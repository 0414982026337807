#!/bin/bash
# more seeds of the same fuzzers on the final tree (appended to gpurun_out/r03_fuzz_more.txt)
set -e
OUT=gpurun_out/r03_fuzz_more.txt
: > $OUT
echo "tree: $(cat .git_head 2>/dev/null || echo unknown)" >> $OUT
for seed in 401 402; do timeout -k 10 280 python3 scratch/fuzz.py $seed 30000 >> $OUT 2>&1; done
echo "--- SW_HIP_LANES=1" >> $OUT
for seed in 411 412; do SW_HIP_LANES=1 timeout -k 10 280 python3 scratch/fuzz.py $seed 30000 >> $OUT 2>&1; done
for seed in 431 432 433; do timeout -k 10 170 python3 scratch/fuzz_package.py $seed 40 >> $OUT 2>&1; done
echo "--- SW_HIP_PACKAGE_LANES=8" >> $OUT
SW_HIP_PACKAGE_LANES=8 timeout -k 10 170 python3 scratch/fuzz_package.py 434 40 >> $OUT 2>&1
for seed in 441 442; do timeout -k 10 170 python3 scratch/fuzz_db.py $seed 100 >> $OUT 2>&1; done
timeout -k 10 170 python3 scratch/fuzz_long.py 421 600 >> $OUT 2>&1
echo "all fuzzers done" >> $OUT
