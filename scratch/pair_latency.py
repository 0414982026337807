"""Latency of the drop-in per-pair call (find_alignment_low_memory) from one thread, by pair size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dlsa_amd
from dlsa_amd import bridge

L = bridge.lib()
sc = dlsa_amd.AlignmentScore(2, -1, -1).c()
rng = np.random.default_rng(1)
for n in (150, 500, 1000, 2000, 4000):
    q = bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8))
    t = bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8))
    for _ in range(20):
        L.free_alignment_result(L.find_alignment_low_memory(q, t, sc))
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        r = L.find_alignment_low_memory(q, t, sc)
        L.free_alignment_result(r)
    dt = (time.perf_counter() - t0) / reps
    st = bridge.last_stats()
    print(f"{n:5d} x {n:5d}: {dt*1e6:8.1f} us per call   exact+walk kernels {st['exact_kernel_ms']*1e3:7.1f} us   {n*n/dt/1e9:6.2f} GCUPS", flush=True)
