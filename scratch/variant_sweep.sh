#!/bin/bash
# GPU box: exact-kernel shapes for each library variant under scratch/variants
for f in scratch/variants/*.so default; do
  echo "== $f"
  if [ "$f" = default ]; then unset SW_HIP_LIB; else export SW_HIP_LIB=$PWD/$f; fi
  timeout -k 10 200 python3 scratch/sweep.py c1 c4m b2 b5 2>&1 | grep case | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['case'][:40], d.get('kernels_ms', d.get('stats',{}).get('exact_kernel_ms')), d.get('ms', d.get('search_ms', d.get('s'))))"
done
