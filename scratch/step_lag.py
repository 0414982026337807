#!/usr/bin/env python3
"""Step time and per-strip lag of the exact kernel: one pair of Q x T with 1 strip and with 11 strips (fill only:
score-only entry is not exposed, so the walker is included; a mutated copy keeps its share small).
usage: step_lag.py [rpl]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dlsa_amd
from dlsa_amd import bridge
rng = np.random.default_rng(5)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
T = 100000
t = lut[rng.integers(0, 4, size=T, dtype=np.uint8)]
sc = dlsa_amd.AlignmentScore(2, -1, -1)
res = {}
for Q in (100, 1400):
    q = lut[rng.integers(0, 4, size=Q, dtype=np.uint8)]
    best = 1e9
    for _ in range(4):
        k0 = bridge.last_stats()["exact_kernel_ms"]
        r = bridge.align_batch([q.tobytes()], [t.tobytes()], sc, 2)
        k1 = bridge.last_stats()["exact_kernel_ms"]
        best = min(best, k1 - k0)
    res[Q] = best
    print(f"Q {Q} x T {T}: exact + walk kernels {best:.3f} ms")
