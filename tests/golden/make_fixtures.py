"""Generates tests/golden/oracle_fixtures.json: seeded inputs with the FULL expected result tuple (aligned
strings, score, max_x, max_y) for the three exported entry points, produced by the oracle
(oracle/sw_oracle.c, the line-by-line restatement of the reference).  The reference's own tests assert only
the aligned strings (rust/src/lib.rs:266-268); these fixtures add what they leave unpinned: scores, end
coordinates, the no-hit convention, the padded-length quirks, the ring path proper (T >= 2Q with the lane
phase and ring wrap-around) and equal-sequence inputs (last row).

Run in the BUILD container only (python tests/golden/make_fixtures.py).  When /root/reference is present,
every positive-score skewed-path result is also cross-checked with the reference's own, independent result
checker master/utils/verify.py (verify_score / verify_exists are imported from the reference tree; nothing
of it is copied).  The GPU box only ever sees the JSON.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as o  # noqa: E402

verify = None
if os.path.isdir("/root/reference/master"):
    sys.path.insert(0, "/root/reference")
    try:
        from master.utils import verify as _v
        verify = _v
    except Exception as e:  # an ordinary import error: fixtures are still written, just not cross-checked
        print("reference verifier not importable:", e)

rng = random.Random(0xD15A)


def rnd(n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def mutate(s, rate):
    out = []
    for c in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice("ACGT"))
        out.append(c if x >= rate else rng.choice("ACGT"))
    return "".join(out)


cases = []


def add(tag, q, t, sc=(-1, 2, -1)):
    cases.append(dict(tag=tag, query=q, target=t, scores=dict(gap=sc[0], match=sc[1], miss=sc[2])))


# full path, T % 64 == 0 (quirk Q1), with and without Tp >= Qp
for T in (64, 128, 192, 256):
    for Q in (40, 100, 200):
        q = rnd(Q)
        t = rnd(T - 12) + q[:12]
        add(f"Q1 hit ends on last column Q={Q} T={T}", q, t)
add("Q1 only H(1,T) scores, Tp >= Qp", "A" + "G" * 20, "C" * 63 + "A")
add("Q1 only H(1,T) would score, Tp < Qp", "A" + "G" * 80, "C" * 63 + "A")
# ring path with T < Qp (quirk Q2 + two-phase discovery order)
for _ in range(12):
    Q = rng.randint(1, 31)
    T = rng.randint(2 * Q, 63)
    add("Q2 tiny ring", rnd(Q, "AC"), rnd(T, "AC"))
# ring path proper: lane phase, T > data_store_height (wrap-around), hits near both ends
for Q, T in ((5, 80), (33, 700), (64, 1500), (100, 2100), (150, 5000)):
    q = rnd(Q)
    t = rnd(T)
    p = rng.randint(0, T - Q)
    t = t[:p] + mutate(q, 0.1) + t[p + Q:]
    add(f"ring Q={Q} T={T}", q, t[:T])
    add(f"ring hit at the end Q={Q} T={T}", q, rnd(T - Q) + q)
# equal sequences (one traceback per row in the reference), no hit, empties
s = rnd(90)
add("equal", s, s)
add("equal x3", s, s + s + s)
add("no hit", "AAAA", "TTTT")
add("empty query", "", "ACGT")
add("empty target", "ACGT", "")
# other scoring triples, arbitrary alphabets, several strips
for sc in ((-3, 5, -2), (-2, 1, -1), (-1, 3, -1), (-5, 4, -1)):
    q = rnd(70)
    add(f"scores {sc}", q, rnd(40) + mutate(q, 0.15) + rnd(30), sc)
add("arbitrary alphabet", "the quick brown fox", "a quick brown dog jumps over the quick brown fox again")
q = rnd(600)
add("three strips", q, rnd(100) + mutate(q[50:550], 0.08) + rnd(150))
add("ties: many equal maxima", "ACAC" * 10, "ACAC" * 30)

fixtures = []
for c in cases:
    s = c["scores"]
    exp = {}
    for func in ("lowmem", "simd", "straight"):
        r = o.align(func, c["query"], c["target"], s["gap"], s["match"], s["miss"])
        assert r is not None, c["tag"]
        exp[func] = dict(q_aln=r["q"], t_aln=r["t"], score=r["score"], max_x=r["max_x"], max_y=r["max_y"])
        if verify is not None and func != "straight" and r["score"] > 0:
            assert verify.verify_score(r["q"], r["t"], len(r["q"]), r["score"], s["match"], -s["miss"], -s["gap"]), c["tag"]
            assert verify.verify_exists(c["query"], r["q"], r["max_x"]), c["tag"]
            assert verify.verify_exists(c["target"], r["t"], r["max_y"]), c["tag"]
    fixtures.append(dict(c, expected=exp))

with open(os.path.join(HERE, "oracle_fixtures.json"), "w") as f:
    json.dump(dict(generator="tests/golden/make_fixtures.py", cross_checked_with_reference_verifier=verify is not None,
                   cases=fixtures), f, indent=0)
print(len(fixtures), "fixtures; reference verifier used:", verify is not None)
