#!/usr/bin/env python3
"""Opt-in fuzz on the GPU box, the database path: random databases (mixed lengths, uniform lengths whose last strip
holds <= 8 columns, multiples of 64) and batches of short queries through sw_db_search_topk_batch (queries sharing
score passes) and sw_db_scores, against the oracle.  Usage: python3 scratch/fuzz_db.py [seed] [n_databases]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dlsa_amd
from dlsa_amd import bridge
import oracle_lib

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = random.Random(seed)
SCORES = [(-1, 2, -1), (-1, 3, -1), (-3, 5, -2), (-2, 1, -1), (-2, 3, -3)]
def rnd(k): return "".join(rng.choice("ACGT") for _ in range(k))

t0 = time.time(); bad = 0; done = 0; launches = 0; nq_total = 0
for it in range(n):
    gap, match, miss = rng.choice(SCORES)
    s = dlsa_amd.AlignmentScore(match, miss, gap)
    shape = rng.random()
    nt = rng.randint(130, 500)
    if shape < 0.3: lens = [rng.randint(1, 700) for _ in range(nt)]
    elif shape < 0.6: L = rng.choice([150, 136, 100, 72, 1000 // 4]); lens = [L] * nt
    elif shape < 0.8: lens = [rng.choice([64, 128, 192, 256, 320]) for _ in range(nt)]
    else: lens = [rng.randint(140, 152) for _ in range(nt)]
    targets = [rnd(l) for l in lens]
    nq = rng.randint(2, 9)
    base = rng.randint(20, 300)
    queries = [rnd(base if rng.random() < 0.5 else rng.randint(20, 400)) for _ in range(nq)]
    for q in queries[:3]:
        i = rng.randrange(nt)
        frag = q[rng.randint(0, len(q) // 2):]
        targets[i] = (frag + targets[i])[:len(targets[i])]
    db = bridge.Database(seqs=targets)
    got = db.search_topk_batch(queries, s, 5)
    launches += bridge.last_stats()["score_launches"]; nq_total += nq
    for q, hits in zip(queries, got):
        exp = [oracle_lib.align_with_fallback(q, t, gap, match, miss) for t in targets]
        order = sorted(range(nt), key=lambda i: (-(exp[i]["score"] if exp[i] else 0), i))[:5]
        order = [i for i in order if exp[i] and exp[i]["score"] > 0]
        g = [(h["target_index"], h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"]) for h in hits if h["score"] > 0]
        e = [(i, exp[i]["score"], exp[i]["max_x"], exp[i]["max_y"], exp[i]["q"], exp[i]["t"]) for i in order]
        done += 1
        if g != e:
            bad += 1
            if bad < 5: print("MISMATCH", (gap, match, miss), len(q), sorted(set(lens))[:6], [x[:4] for x in g], [x[:4] for x in e])
    db.close()
print("fuzz_db seed", seed, "searches", done, "score launches", launches, "for", nq_total, "queries, mismatches", bad, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
