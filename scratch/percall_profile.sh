#!/bin/bash
# GPU box: kernel-level breakdown of single per-pair calls through the C harness
python3 - <<PY
import random
rng=random.Random(1)
r=lambda n:"".join(rng.choice("ACGT") for _ in range(n))
q=r(1000)
with open("/tmp/p1k.txt","w") as f:
    for i in range(300): f.write(f"-1 2 -1 {q} {r(rng.randint(900,1100))}\n")
with open("/tmp/p150.txt","w") as f:
    for i in range(300): f.write(f"-1 2 -1 {r(150)} {r(150)}\n")
PY
export TMPDIR=/tmp
for f in p1k p150; do
  timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$f -o s -- ./tests/c_harness/worker_harness pairs /tmp/$f.txt 1 > /dev/null 2> gpurun_out/h_$f.log
  grep pairs gpurun_out/h_$f.log | tail -1
  cut -d, -f1-4 gpurun_out/prof_$f/s_kernel_stats.csv | cut -c1-120
done
