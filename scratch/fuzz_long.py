#!/usr/bin/env python3
"""Opt-in fuzz on the GPU box, long pairs: mutated copies (substitutions and indels) and unrelated sequences of
1-7 kb through sw_align_batch on all three entry semantics, singly (strips of one pair chained through HBM, the
walker's tiles and window changes) and as one batch, against the oracle.  Usage: python3 scratch/fuzz_long.py [seed] [n]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dlsa_amd
from dlsa_amd import bridge
import oracle_lib

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = random.Random(seed)
SCORES = [(-1, 2, -1), (-3, 5, -2), (-2, 1, -1), (-1, 1, -1), (-2, 3, -3)]
FUNC = {0: "lowmem", 1: "simd", 2: "straight"}

def rnd(k): return "".join(rng.choice("ACGT") for _ in range(k))

def mutate(s, p_sub, p_indel):
    out = []
    for c in s:
        r = rng.random()
        if r < p_indel / 2: continue
        if r < p_indel: out.append(rng.choice("ACGT"))
        out.append(rng.choice("ACGT") if rng.random() < p_sub else c)
    return "".join(out)

def gen_pair():
    Q = rng.randint(1000, 7000)
    q = rnd(Q)
    mode = rng.random()
    if mode < 0.3: t = rnd(rng.randint(1000, 7000))
    elif mode < 0.8: t = rnd(rng.randint(0, 500)) + mutate(q[rng.randint(0, Q // 3):], rng.choice([0.02, 0.1, 0.25]), rng.choice([0.0, 0.03, 0.15])) + rnd(rng.randint(0, 500))
    else: t = mutate(q, 0.05, 0.05)[:rng.randint(500, Q)]
    return q, t

t0 = time.time(); bad = 0; done = 0
for it in range(n):
    path = rng.randrange(3); sc = rng.choice(SCORES)
    gap, match, miss = sc
    pairs = [gen_pair() for _ in range(rng.choice([1, 1, 4]))]
    got = bridge.align_batch([p[0] for p in pairs], [p[1] for p in pairs], dlsa_amd.AlignmentScore(match, miss, gap), path)
    for (q, t), g in zip(pairs, got):
        e = oracle_lib.align(FUNC[path], q, t, gap, match, miss)
        done += 1
        if e is None:
            ok = g is None
        elif e["score"] > 32767:
            continue  # (outside the reference's i16 domain: the oracle wraps)
        else:
            ok = g is not None and (g.query, g.target, g.score, g.max_x, g.max_y) == (e["q"], e["t"], e["score"] & 0xFFFF, e["max_x"], e["max_y"])
        if not ok:
            bad += 1
            if bad < 5: print("MISMATCH", path, sc, len(q), len(t), g and (g.score, g.max_x, g.max_y, len(g.query)), e and (e["score"], e["max_x"], e["max_y"], len(e["q"])))
print("fuzz_long seed", seed, "pairs", done, "mismatches", bad, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
