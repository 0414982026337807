#!/bin/bash
cd "$(dirname "$0")/.."
for shape in gen:10000:100000:1000 gen:10000:100000:1030 gen:10000:200000:520 gen:10000:100000:1024 gen:10000:60000:1500 gen:10000:40000:2600 gen:1000:100000:1030 c3 c5s c1big; do
  timeout -k 10 200 python3 scratch/sweep.py $shape | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d['stats']
print('%-44s kernel_ms %8.2f  gcups %8.1f  per_Mrow_ms %.4f' % (d['case'][:44], s['score_kernel_ms'], d['gcups'], s['score_kernel_ms'] / max(s['score_strip_rows'],1) * 1e6))" || exit 1
done
