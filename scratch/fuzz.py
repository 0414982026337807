#!/usr/bin/env python3
"""Opt-in fuzz on the GPU box: many random pairs through sw_align_batch (all three entry semantics) and through the
database path, against the oracle.  Usage: python3 scratch/fuzz.py [seed] [n_pairs]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dlsa_amd
from dlsa_amd import bridge
import oracle_lib

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rng = random.Random(seed)
SCORES = [(-1, 2, -1), (-1, 3, -1), (-3, 5, -2), (-2, 1, -1), (-1, 1, -1), (-2, 3, -3), (-5, 4, -1), (-1, 2, -4), (-7, 9, -1)]
FUNC = {0: "lowmem", 1: "simd", 2: "straight"}

def rnd(k, alpha): return "".join(rng.choice(alpha) for _ in range(k))

def gen_pair():
    alpha = rng.choice(["ACGT", "ACGT", "AC", "A", "ACGTN", "xyzé", "ACGU"])
    mode = rng.random()
    if mode < 0.25: Q, T = rng.randint(0, 40), rng.randint(0, 130)
    elif mode < 0.4: Q = rng.randint(1, 31); T = rng.randint(2 * Q, 70)
    elif mode < 0.7: Q = rng.randint(30, 700); T = rng.choice([64, 128, 192, 256, 320, 640, rng.randint(1, 900)])
    elif mode < 0.9: Q = rng.randint(1, 300); T = rng.randint(2 * Q, 2 * Q + 1200)
    else: Q = rng.randint(500, 1500); T = rng.randint(300, 2500)
    q, t = rnd(Q, alpha), rnd(T, alpha)
    if rng.random() < 0.5 and Q > 3 and T > 3:
        p = rng.randint(0, T - 1); frag = q[rng.randint(0, Q // 2):rng.randint(Q // 2, Q)]
        frag = "".join(c if rng.random() > 0.1 else rng.choice(alpha) for c in frag)
        t = (t[:p] + frag + t[p:])[:T]
    return q, t

t0 = time.time(); bad = 0
for path in (0, 1, 2):
    for sc in SCORES:
        m = max(1, n // (3 * len(SCORES)))
        pairs = [gen_pair() for _ in range(m)]
        gap, match, miss = sc
        got = bridge.align_batch([p[0] for p in pairs], [p[1] for p in pairs], dlsa_amd.AlignmentScore(match, miss, gap), path)
        for (q, t), g in zip(pairs, got):
            e = oracle_lib.align(FUNC[path], q, t, gap, match, miss)
            if e is None:
                ok = g is None
            else:
                ok = g is not None and (g.query, g.target, g.score, g.max_x, g.max_y) == (e["q"], e["t"], e["score"] & 0xFFFF, e["max_x"], e["max_y"])
            if not ok:
                bad += 1
                if bad < 5: print("MISMATCH", path, sc, len(q), len(t), repr(q[:60]), repr(t[:60]), g, e)
# database path
for sc in SCORES[:5]:
    gap, match, miss = sc
    q = rnd(rng.randint(50, 400), "ACGT")
    targets = [gen_pair()[1] for _ in range(400)]
    targets = [t for t in targets if "\x00" not in t]
    db = bridge.Database(seqs=targets)
    got = db.scores(q, dlsa_amd.AlignmentScore(match, miss, gap))
    for i, t in enumerate(targets):
        e = oracle_lib.align_with_fallback(q, t, gap, match, miss)
        if e is None or int(got[i]) != e["score"]:
            bad += 1
            if bad < 5: print("DB MISMATCH", sc, len(q), len(t), int(got[i]), e and e["score"])
    db.close()
print("fuzz seed", seed, "pairs", n, "mismatches", bad, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
