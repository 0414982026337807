#!/bin/bash
# score kernel time against the maximal lane length (units) of the ticket order, for target lengths around strip multiples
cd "$(dirname "$0")/.."
for shape in gen:10000:100000:1030 gen:10000:200000:520 gen:10000:100000:1000; do
  for lmax in 1 6 10 14 24 70; do
    echo -n "$shape LMAX=$lmax  "
    SW_HIP_SCORE_LMAX=$lmax timeout -k 10 120 python3 scratch/sweep.py $shape | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d['stats']
print('kernel_ms %.2f strip_rows %d chain_rows %d per_Mrow_ms %.4f' % (s['score_kernel_ms'], s['score_strip_rows'], s['score_chain_rows'], s['score_kernel_ms'] / s['score_strip_rows'] * 1e6))" || exit 1
  done
done
