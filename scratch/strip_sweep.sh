#!/bin/bash
# GPU box: score kernel with 1, 2, 4, 16 strips per group at constant cell count (isolates the wave pipeline)
for tl in 16 32 64 256; do
  n=$((100000000/tl))
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --tlen $tl --targets $n > /tmp/ss.json
  python3 - <<PY
import json
d=json.loads(open("/tmp/ss.json").read().strip().split("\n")[-1])
k=d["roofline"]["kernel_ms"]
print("tlen", $tl, "targets", $n, "kernel_ms", round(k,2), "TCUPS", round(d["config"]["cells_per_gpu_per_step"]/k/1e9,2))
PY
done
