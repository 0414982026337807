#!/usr/bin/env python3
"""Opt-in fuzz on the GPU box, the work-package fast path: packages of a few queries x a few dozen long targets
(homologs planted, with substitutions and indels, also near chunk and unit boundaries) through sw_align_batch against
the oracle, every pair.  Usage: python3 scratch/fuzz_package.py [seed] [n_packages] [sets]
With `sets`: every pair of a batch has a query and a target of its own (what a combined batch of per-pair callers looks
like): as many target sets of one target, which run side by side on the context lanes."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dlsa_amd
from dlsa_amd import bridge
import oracle_lib

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = random.Random(seed)
SCORES = [(-1, 2, -1), (-3, 5, -2), (-2, 1, -1), (-2, 3, -3), (-4, 2, -3)]

def rnd(k): return "".join(rng.choice("ACGT") for _ in range(k))
def mutate(s, p_sub, p_indel):
    out = []
    for c in s:
        r = rng.random()
        if r < p_indel / 2: continue
        if r < p_indel: out.append(rng.choice("ACGT"))
        out.append(rng.choice("ACGT") if rng.random() < p_sub else c)
    return "".join(out)

t0 = time.time(); bad = 0; done = 0; fast = 0
for it in range(n):
    gap, match, miss = rng.choice(SCORES)
    nq, nt = rng.randint(1, 5), rng.randint(8, 40)
    qs = [rnd(rng.randint(60, 900)) for _ in range(nq)]
    ts = []
    for _ in range(nt):
        T = rng.randint(2500, 40000)
        t = rnd(T)
        for _k in range(rng.randint(0, 3)):
            q = rng.choice(qs)
            frag = mutate(q[rng.randint(0, len(q) // 3):], rng.choice([0.0, 0.05, 0.2]), rng.choice([0.0, 0.02, 0.1]))
            p = rng.choice([0, max(0, T - len(frag)), rng.randint(0, T - 1), (rng.randint(1, max(1, T // 256)) * 256) % T])
            t = (t[:p] + frag + t[p + len(frag):])[:T]
        ts.append(t)
    if len(sys.argv) > 3 and sys.argv[3] == "sets":
        Qs = [rng.choice(qs) if rng.random() < 0.3 else mutate(rng.choice(qs), 0.3, 0.1) + rnd(rng.randint(40, 200)) for _ in ts]
        Qs = [q for q in Qs if len(q) >= 40] + qs[:max(0, len(ts) - len(Qs))]
        Qs = (Qs * 2)[:len(ts)]
        Qs = [str(q) + "" for q in Qs]  # (distinct objects: a query of its own per pair)
        Ts = ts
    else:
        Qs = [q for q in qs for _ in ts]; Ts = ts * nq
    got = bridge.align_batch(Qs, Ts, dlsa_amd.AlignmentScore(match, miss, gap), 0)
    fast += bridge.package_stats()["pairs"]
    for q, t, g in zip(Qs, Ts, got):
        e = oracle_lib.align("lowmem", q, t, gap, match, miss)
        done += 1
        if e is None: ok = g is None
        elif e["score"] > 32767: continue
        else: ok = g is not None and (g.query, g.target, g.score, g.max_x, g.max_y) == (e["q"], e["t"], e["score"] & 0xFFFF, e["max_x"], e["max_y"])
        if not ok:
            bad += 1
            if bad < 5: print("MISMATCH", (gap, match, miss), len(q), len(t), g and (g.score, g.max_x, g.max_y, len(g.query)), e and (e["score"], e["max_x"], e["max_y"], len(e["q"])))
print("fuzz_package seed", seed, "pairs", done, "through the fast path", fast, "mismatches", bad, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
