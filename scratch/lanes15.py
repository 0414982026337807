#!/usr/bin/env python3
"""15 pairs of the reference's synthetic default (each with a target of its own) in one sw_align_batch: what a combined
batch of 15 per-pair callers is.  SW_HIP_PACKAGE_LANES=1 runs the target sets one after the other."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dlsa_amd
from dlsa_amd import bridge
rng = random.Random(5)
r = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15
qs = [r(rng.randint(200, 1000)) for _ in range(n)]
ts = [r(rng.randint(10000, 200000)) for _ in range(n)]
sc = dlsa_amd.AlignmentScore(2, -1, -1)
for rep in range(4):
    t0 = time.perf_counter(); res = bridge.align_batch(qs, ts, sc, 0); dt = time.perf_counter() - t0
    print(f"rep {rep}: {dt*1e3:.2f} ms for {n} pairs, check {sum(x.score for x in res)}", flush=True)
