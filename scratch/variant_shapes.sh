#!/bin/bash
# GPU box: the database shapes of scratch/sweep.py for each library variant under scratch/variants, then the default
for f in scratch/variants/*.so default; do
  echo "== $f"
  if [ "$f" = default ]; then unset SW_HIP_LIB; else export SW_HIP_LIB=$PWD/$f; fi
  timeout -k 10 300 python3 scratch/sweep.py c1 c1big c2 c3 c5s 2>&1 | grep case | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['case'][:44].ljust(44), 'kernel %.3f ms' % d['stats']['score_kernel_ms'], 'search %.3f ms' % d['search_ms'], 'top', d['top'])" || exit 1
done
