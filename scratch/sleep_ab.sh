#!/bin/bash
# same box: back-off of the LDS flag waits of the score kernel (units of 64 clocks)
set -e
for v in base ls4 ls8 ls32 r8d16 r16d8 base; do
  echo "== $v"
  SW_HIP_LIB=$PWD/scratch/variants/libsw_$v.so timeout -k 10 200 python3 scratch/sweep.py c2 c3 c5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  %-45s kernel %8.3f ms' % (d['case'], d['stats']['score_kernel_ms']))
"
done
