"""The "clean" int32 model (oracle/sw_clean.c -- the semantics the HIP kernels implement, DESIGN.md
"Effective semantics") equals the line-by-line restatement of the reference (oracle/sw_oracle.c) on the
i16-safe domain, including the reference's quirks.  CPU only."""
import random

import pytest

SCORES = [(-1, 2, -1), (-1, 3, -1), (-3, 5, -2), (-2, 1, -1), (-1, 1, -1), (-2, 3, -3), (-5, 4, -1)]
PATHS = ((0, "lowmem"), (1, "simd"), (2, "straight"))


def _rnd(rng, n, alpha):
    return "".join(rng.choice(alpha) for _ in range(n))


def _same(o, q, t, sc):
    for path, f in PATHS:
        a = o.align(f, q, t, *sc)
        c = o.clean_align(path, q, t, *sc)
        assert a is not None and c is not None
        exp = (c["q"], c["t"], c["score"] if path != 2 else 0, c["max_x"], c["max_y"])
        assert (a["q"], a["t"], a["score"], a["max_x"], a["max_y"]) == exp, (f, q, t, sc)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_clean_equals_restatement_random(oracle, seed):
    rng = random.Random(seed)
    for _ in range(500):
        mode = rng.random()
        alpha = rng.choice(["AC", "ACGT", "ACGT", "A"])
        if mode < 0.35:
            Q, T = rng.randint(0, 40), rng.randint(0, 100)
        elif mode < 0.6:  # ring path with T < Qp (quirk Q2 + two-phase discovery order)
            Q = rng.randint(1, 31)
            T = rng.randint(2 * Q, 70)
        elif mode < 0.8:  # full path, T % 64 == 0 (quirk Q1)
            Q = rng.randint(30, 200)
            T = rng.choice([64, 128, 192, 256, rng.randint(1, 400)])
        else:  # ring path proper: lane phase + ring wrap-around
            Q = rng.randint(1, 100)
            T = rng.randint(2 * Q, 2 * Q + 600)
        q, t = _rnd(rng, Q, alpha), _rnd(rng, T, alpha)
        if rng.random() < 0.3 and Q > 3 and T > Q:
            p = rng.randint(0, T - 1)
            t = (t[:p] + q[:rng.randint(1, Q)] + t[p:])[:T]
        _same(oracle, q, t, rng.choice(SCORES))


def test_quirk_q1_last_column_dropped(oracle):
    # T % 64 == 0 on the full path: the last target character is not aligned (except H(1,T))
    q = "GGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGTTTTTTTTTACGTA"
    t = "C" * 59 + "ACGTA"  # T = 64; the match "ACGTA" ends on the last column
    a = oracle.align("simd", q, t, -1, 2, -1)
    assert a["score"] == 8 and a["q"] == "ACGT"  # not 10 / "ACGTA"
    assert oracle.clean_score_plain(q, t, -1, 2, -1) == 10
    _same(oracle, q, t, (-1, 2, -1))
    # H(1,T) is still computed when Tp >= Qp: a query whose FIRST char matches the LAST target char
    q2 = "A" + "G" * 20
    t2 = "C" * 63 + "A"
    a2 = oracle.align("simd", q2, t2, -1, 2, -1)
    assert (a2["q"], a2["t"], a2["score"], a2["max_x"], a2["max_y"]) == ("A", "A", 2, 0, 63)
    _same(oracle, q2, t2, (-1, 2, -1))
    # ... and is NOT when Tp < Qp
    q3 = "A" + "G" * 80
    a3 = oracle.align("simd", q3, t2, -1, 2, -1)
    assert (a3["q"], a3["score"], a3["max_x"]) == ("", 0, oracle.U64MAX)
    _same(oracle, q3, t2, (-1, 2, -1))


def test_ring_wraparound_and_equal_sequences(oracle):
    rng = random.Random(5)
    # T > data_store_height forces the ring to wrap (algorithm/mod.rs:240)
    q = _rnd(rng, 40, "ACGT")
    t = _rnd(rng, 700, "ACGT") + q[5:35] + _rnd(rng, 300, "ACGT")
    _same(oracle, q, t, (-1, 2, -1))
    # equal prefixes: one traceback per row (report 4.3.1 worst case)
    q = _rnd(rng, 70, "ACGT")
    _same(oracle, q, q + q + q, (-1, 2, -1))
    _same(oracle, q, q, (-1, 2, -1))
