#!/usr/bin/env python3
"""Per-strip accumulators of the SW_EXACT_TIMELINE build of sw_exact_kernel after ONE pair of Q x T (straight path):
time of the strip, time in the boundary poll (incl. the wait for the fetched words), spins, blocks that found their
boundary missing, time from the end of a block's arithmetic to the end of its stores' issue.
usage: exact_timeline.py Q T"""
import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dlsa_amd
from dlsa_amd import bridge
Q, T = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(5)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
q = lut[rng.integers(0, 4, size=Q, dtype=np.uint8)].tobytes()
t = lut[rng.integers(0, 4, size=T, dtype=np.uint8)].tobytes()
sc = dlsa_amd.AlignmentScore(2, -1, -1)
lib = ctypes.CDLL(os.environ["SW_HIP_LIB"])
lib.sw_hip_debug_exact_timeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for _ in range(2):
    k0 = bridge.last_stats()["exact_kernel_ms"]
    bridge.align_batch([q], [t], sc, 2)
    kms = bridge.last_stats()["exact_kernel_ms"] - k0
buf = np.zeros(4096 * 8, dtype=np.uint32)
assert lib.sw_hip_debug_exact_timeline(buf.ctypes.data, buf.size) == 0
A = buf.reshape(4096, 8)
n = int((A[:, 5] > 0).sum())
print(f"Q {Q} x T {T}: kernels {kms:.3f} ms, {n} strips, {int(A[0,5])} blocks of 16 steps each")
print("strip  total_us  per_step_ns  poll_us  spins  GHz  stores_us  compute_us  compute_ns_per_step")
b0 = int(A[0, 6])
for s in list(range(min(n, 6))) + list(range(6, n, max(1, n // 8))):
    tot, poll, spins, cyc16, store, ng, beg, comp = [int(x) for x in A[s, :8]]
    print(f"{s:5d} {tot/100:9.1f} {tot*10/(ng*16):12.1f} {poll/100:8.1f} {spins:6d} {cyc16*16/(tot*10):6.2f} {store/100:10.1f} {comp/100:10.1f} {comp*10/(ng*16):8.1f}")
