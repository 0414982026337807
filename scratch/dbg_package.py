import random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import dlsa_amd, oracle_lib as oracle
from dlsa_amd import bridge
SC = (-1, 2, -1)
def _rnd(rng, n, alpha="ACGT"): return "".join(rng.choice(alpha) for _ in range(n))
def _mut(rng, s, rate): return "".join((rng.choice("ACGT") if rng.random() < rate else c) for c in s)
rng = random.Random(41)
s = dlsa_amd.AlignmentScore(SC[1], SC[2], SC[0])
queries = [_rnd(rng, n) for n in (200, 517, 1000, 64)]
targets = [_rnd(rng, rng.randint(3000, 9000)) for _ in range(20)] + [_rnd(rng, 20000), _rnd(rng, 4096), _rnd(rng, 2048)]
targets[0] = targets[0][:1000] + _mut(rng, queries[2][50:950], 0.04) + targets[0][1000:]
targets[1] = targets[1][:250] + queries[1][10:500] + targets[1][250:]
frag = queries[0][20:180]
targets[2] = targets[2][:300] + frag + targets[2][300:2000] + frag + targets[2][2000:]
targets[3] = targets[3] + queries[0][:150]
targets[4] = queries[1][100:400] + targets[4]
qs = [q for q in queries for _ in targets]
ts = targets * len(queries)
for qi, q in enumerate(queries):
    res = bridge.align_batch([q] * len(targets), targets, s, 0)
    st = bridge.package_stats()
    print("query", qi, len(q), st)
    for ti, (r, t) in enumerate(zip(res, targets)):
        e = oracle.align("lowmem", q, t, *SC)
        ok = (r.score, r.max_x, r.max_y, r.query, r.target) == (e["score"], e["max_x"], e["max_y"], e["q"], e["t"])
        if not ok:
            print("  MISMATCH target", ti, len(t), (r.score, r.max_x, r.max_y, len(r.query)), (e["score"], e["max_x"], e["max_y"], len(e["q"])))
print("---- divergence analysis")
q = queries[2]
for ti in (5, 12, 19):
    t = targets[ti]
    r = bridge.align_batch([q], [t], s, 0)[0]
    print(bridge.package_stats())
    e = oracle.align("lowmem", q, t, *SC)
    a, b = (r.query, r.target), (e["q"], e["t"])
    # walk from the end
    x, j = r.max_x, r.max_y
    k = 0
    while k < min(len(a[0]), len(b[0])) and a[0][-1 - k] == b[0][-1 - k] and a[1][-1 - k] == b[1][-1 - k]:
        if a[0][-1 - k] != '-': x -= 1
        if a[1][-1 - k] != '-': j -= 1
        k += 1
    print("target", ti, "diverge after", k, "columns from the end at cell x=%d j=%d (0-based next), j%%64=%d x%%256=%d" % (x, j, (j + 1) % 64, (x + 1) % 256),
          "gpu:", a[0][-1 - k:][:1], a[1][-1 - k:][:1], "ref:", b[0][-1 - k:][:1], b[1][-1 - k:][:1])
