#!/bin/bash
# usage: scratch/harness_bench.sh  (GPU box): drop-in call pattern, one pair per call, 1 vs 15 threads
python3 - <<PY
import random
rng=random.Random(1)
r=lambda n:"".join(rng.choice("ACGT") for _ in range(n))
q=r(1000)
with open("/tmp/pairs_1k.txt","w") as f:
    for i in range(3000): f.write(f"-1 2 -1 {q} {r(rng.randint(900,1100))}\n")
with open("/tmp/pairs_150.txt","w") as f:
    for i in range(20000): f.write(f"-1 2 -1 {r(150)} {r(150)}\n")
PY
for t in 1 15 64; do
  for f in pairs_1k pairs_150; do
    echo -n "$f: "; ./tests/c_harness/worker_harness pairs /tmp/$f.txt $t 2>&1 > /tmp/out_${f}_$t.txt
  done
done
