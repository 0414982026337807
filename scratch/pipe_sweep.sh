#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2; do
for parts in off 2 3 4 6; do
  if [ $parts = off ]; then export SW_HIP_PIPE_MIN_PAIRS=100000000; unset SW_HIP_PIPE_PARTS; else unset SW_HIP_PIPE_MIN_PAIRS; export SW_HIP_PIPE_PARTS=$parts; fi
  echo -n "parts=$parts  "
  timeout -k 10 200 python3 scratch/sweep.py w2 w5 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    d = json.loads(line); print('%s: %.2f + %.2f ms   ' % (d['case'][16:34], d['call_ms'], d['free_ms']), end='')
print()" || exit 1
done
done
