#!/usr/bin/env python3
"""Repeated calls of the round-3 paths: resident memory must level off (tables, slabs and arenas are handed from call to
call, nothing may pile up)."""
import os, sys, time, random, resource
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import dlsa_amd
from dlsa_amd import bridge

def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * 4096 / 1e6

rng = random.Random(3)
r = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
L = bridge.lib()
# (1) sw_execute_work, 100 000 pairs of reads
fq, oq = bridge.synth(11, 1, 150, 150)
ft, ot = bridge.synth(12, 100000, 150, 150)
seqs = [fq[int(oq[0]):int(oq[1])].tobytes()] + [ft[int(ot[i]):int(ot[i + 1])].tobytes() for i in range(100000)]
arr = (C.c_char_p * len(seqs))(*seqs)
n = 100000
pr = np.empty((n, 2), dtype=np.uint32); pr[:, 0] = 0; pr[:, 1] = np.arange(1, n + 1, dtype=np.uint32)
pkg = bridge.CWorkPackage(arr, len(seqs), pr.ctypes.data_as(C.POINTER(bridge.CWorkPair)), n, 2, 1, 1)
out = (bridge.CAlignmentDetail * n)()
for it in range(40):
    rc = L.sw_execute_work(C.byref(pkg), out); assert rc >= 0
    L.sw_alignment_details_free(out, n)
    if it % 10 == 9: print(f"execute_work x{it+1}: rss {rss_mb():.0f} MB", flush=True)
# (2) combined-caller shape: 15 pairs with targets of their own (context lanes)
qs = [r(rng.randint(200, 1000)) for _ in range(15)]
ts = [r(rng.randint(10000, 120000)) for _ in range(15)]
sc = dlsa_amd.AlignmentScore(2, -1, -1)
for it in range(150):
    res = bridge.align_batch(qs, ts, sc, 0)
    del res
    if it % 50 == 49: print(f"lanes batch x{it+1}: rss {rss_mb():.0f} MB", flush=True)
# (3) 8 000 pairs of 1 kb
q1 = [r(1000) for _ in range(8)]
t1 = [r(rng.randint(900, 1100)) for _ in range(1000)]
qs = [q for q in q1 for _ in t1]; ts = t1 * 8
for it in range(20):
    res = bridge.align_batch(qs, ts, sc, 0)
    del res
    if it % 10 == 9: print(f"1 kb batch x{it+1}: rss {rss_mb():.0f} MB", flush=True)
print("soak done")
