#!/bin/bash
cd "$(dirname "$0")/.."
for q in 50 100 150 300 600 1200 2400 4800 9600; do
  timeout -k 10 200 python3 scratch/sweep.py gen:$q:1000000:150 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d['stats']
units = s['score_strip_rows'] / $q / 16
print('Q=%5d kernel_ms %8.3f  tcups %6.2f  per_Mrow_ms %.4f  us_per_unit_per_wg %.2f' % ($q, s['score_kernel_ms'], d['cells']/s['score_kernel_ms']/1e9, s['score_kernel_ms'] / max(s['score_strip_rows'],1) * 1e6, s['score_kernel_ms']*1e3/(units/256)))" || exit 1
done
