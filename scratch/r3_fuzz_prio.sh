#!/bin/bash
# the fuzzers that reach the score kernel (databases, work packages) plus one seed of pairs, on the tree with issue
# priorities and 6-row blocks everywhere; totals appended to profiles/r03_fuzz.txt
set -e
OUT=gpurun_out/r03_fuzz_prio.txt
: > $OUT
echo "tree: $(cat .git_head 2>/dev/null || echo unknown) (score kernel: issue priorities, 6-row blocks for every one-query pass)" >> $OUT
timeout -k 10 170 python3 scratch/fuzz_db.py 541 100 >> $OUT 2>&1
timeout -k 10 170 python3 scratch/fuzz_db.py 542 100 >> $OUT 2>&1
timeout -k 10 170 python3 scratch/fuzz_package.py 531 40 >> $OUT 2>&1
timeout -k 10 170 python3 scratch/fuzz_package.py 532 40 >> $OUT 2>&1
timeout -k 10 170 python3 scratch/fuzz_package.py 551 60 sets >> $OUT 2>&1
timeout -k 10 280 python3 scratch/fuzz.py 501 30000 >> $OUT 2>&1
echo "all fuzzers done" >> $OUT
