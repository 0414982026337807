#!/bin/bash
# GPU box: 150-nt reads (C5) with hand-over blocks of 12 rows (ring of 3 blocks) against 6 rows (ring of 6 blocks)
set -e
cd "$(dirname "$0")/.."
echo "== B 12"; timeout -k 10 200 python3 scratch/sweep.py c5 | grep -o '"search_ms": [0-9.]*\|score_kernel_ms": [0-9.]*'
echo "== B 6 / D 6"; SW_HIP_SCORE_FEW_GROUPS=100000000 timeout -k 10 200 python3 scratch/sweep.py c5 | grep -o '"search_ms": [0-9.]*\|score_kernel_ms": [0-9.]*'
echo "== timeline B 6 / D 6"; SW_HIP_SCORE_FEW_GROUPS=100000000 scratch/score_timeline.sh run 150 10000000 150 | head -4
echo "== small passes with the deeper ring"; timeout -k 10 100 python3 scratch/lanes15.py 1; timeout -k 10 100 python3 scratch/lanes15.py 15; timeout -k 10 100 python3 scratch/bench_rust.py | tail -2; timeout -k 10 100 python3 scratch/sweep.py c1 | grep -o '"search_ms": [0-9.]*'
