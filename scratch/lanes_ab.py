#!/usr/bin/env python3
"""A/B on the GPU box: a batch of pairs through sw_align_batch with the lane kernel forbidden / forced / automatic.
Usage: python3 scratch/lanes_ab.py Q T N [nq]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dlsa_amd
from dlsa_amd import bridge
import ctypes as C
Q, T, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rng = np.random.default_rng(5)
lut = np.frombuffer(b"ATCG", dtype=np.uint8)
qs = [lut[rng.integers(0, 4, size=Q, dtype=np.uint8)].tobytes() for _ in range(nq)]
ts = [lut[rng.integers(0, 4, size=int(rng.integers(int(T * 0.9), int(T * 1.1) + 1)), dtype=np.uint8)].tobytes() for _ in range(N // nq)]
Qs = [q for q in qs for _ in ts]; Ts = ts * nq
n = len(Qs)
L = bridge.lib(); sc = dlsa_amd.AlignmentScore(2, -1, -1)
qa = (C.c_char_p * n)(*Qs); ta = (C.c_char_p * n)(*Ts)
cells = float(sum(len(a) * len(b) for a, b in zip(Qs, Ts)))
for mode in ("0", "1", "auto"):
    os.environ["SW_HIP_LANES"] = mode
    best = 1e9
    for _ in range(3):
        out = (C.POINTER(bridge.CResult) * n)()
        s0 = bridge.last_stats()
        t0 = time.perf_counter(); rc = L.sw_align_batch(qa, ta, n, sc.c(), 0, out); dt = time.perf_counter() - t0
        assert rc == 0, bridge.last_error()
        s1 = bridge.last_stats()
        chk = sum(int(out[i].contents.score) for i in range(0, n, max(1, n // 50)))
        for i in range(n): L.free_alignment_result(out[i])
        if dt < best:
            best = dt; kms = s1["exact_kernel_ms"] - s0["exact_kernel_ms"]; lp = s1["lanes_pairs"] - s0["lanes_pairs"]
            lms = s1["lanes_kernel_ms"] - s0["lanes_kernel_ms"]
    print(json.dumps({"Q": Q, "T": T, "pairs": n, "lanes": mode, "call_ms": round(best * 1e3, 3), "kernels_ms": round(kms, 3),
                      "lanes_kernel_ms": round(lms, 3), "lanes_pairs": lp, "kernel_gcups": round(cells / kms / 1e6, 1), "check": chk}), flush=True)
