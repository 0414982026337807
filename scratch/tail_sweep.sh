#!/bin/bash
# GPU box: does the score kernel's time step with the number of unit rounds? (kernel ms per 1000 strips)
for n in 90112 98304 99840 100000 102400 104448 106496; do
  timeout -k 10 100 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --targets $n > /tmp/ts.json 2>/dev/null
  python3 - <<PY
import json
d=json.loads(open("/tmp/ts.json").read().strip().split("\n")[-1])
n=$n; groups=(n+127)//128; strips=groups*63; units=strips/16
k=d["roofline"]["kernel_ms"]
print(f"targets {n} groups {groups} units {units:.1f} rounds {units/256:.2f} kernel_ms {k:.2f} ms/1000strips {k/strips*1000:.4f}")
PY
done
