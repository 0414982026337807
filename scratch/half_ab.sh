#!/bin/bash
# A/B on one box: score kernel with / without the half-width form of a group's last strip
set -e
for rep in 1 2; do
for v in nohalf half; do
  echo "== $v ($rep)"
  SW_HIP_LIB=$PWD/scratch/variants/libsw_$v.so timeout -k 10 200 python3 scratch/sweep.py c2 c1big c3 c5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  %-45s search %8.3f ms  kernel %8.3f ms' % (d['case'], d['search_ms'], d['stats']['score_kernel_ms']))
"
done
done
