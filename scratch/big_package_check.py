#!/usr/bin/env python3
"""A work package whose database has MORE than 128 lane groups (3 queries x 3000 targets of 20-40 kb, homologs planted)
through the fast path, a sample of its pairs against the oracle."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dlsa_amd
from dlsa_amd import bridge
import oracle_lib
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 9)
def rnd(k): return "".join(rng.choice("ACGT") for _ in range(k))
qs = [rnd(rng.randint(300, 900)) for _ in range(3)]
ts = []
for i in range(3000):
    T = rng.randint(20000, 40000)
    t = rnd(T)
    if i % 7 == 0:
        q = rng.choice(qs); p = rng.randint(0, T - len(q)); frag = q[rng.randint(0, 100):]
        t = (t[:p] + frag + t[p + len(frag):])[:T]
    ts.append(t)
Qs = [q for q in qs for _ in ts]; Ts = ts * len(qs)
t0 = time.time()
got = bridge.align_batch(Qs, Ts, dlsa_amd.AlignmentScore(2, -1, -1), 0)
dt = time.time() - t0
st = bridge.package_stats()
sample = [k for k in range(len(Qs)) if (k % 3000) % 7 == 0][:90] + rng.sample(range(len(Qs)), 60)
bad = 0
for k in sample:
    e = oracle_lib.align("lowmem", Qs[k], Ts[k], -1, 2, -1)
    g = got[k]
    ok = (g is None) if e is None else (g is not None and (g.query, g.target, g.score, g.max_x, g.max_y) == (e["q"], e["t"], e["score"] & 0xFFFF, e["max_x"], e["max_y"]))
    bad += 0 if ok else 1
print(f"big package: {len(Qs)} pairs in {dt:.2f} s, package stats {st}, sample of {len(sample)} pairs against the oracle: {bad} mismatches")
