"""GPU parity of the database path (include/sw_hip.h: sw_db_create / sw_db_scores / sw_db_search_topk):
the inter-task packed-int16 score kernel + re-run rules + top-k + exact alignments, against the oracle's
find_alignment_low_memory restatement on the same seeded inputs."""
import random

import pytest

pytestmark = pytest.mark.gpu


def _rnd(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def _oracle_all(oracle, q, targets, sc):
    return [oracle.align("lowmem", q, t, *sc) for t in targets]


def _topk(exp, k):
    order = sorted(range(len(exp)), key=lambda i: (-exp[i]["score"], i))[:k]
    return order


def _run(oracle, q, targets, sc, k=5):
    import dlsa_amd
    from dlsa_amd import bridge
    gap, match, miss = sc
    s = dlsa_amd.AlignmentScore(match, miss, gap)
    db = bridge.Database(seqs=targets)
    try:
        exp = _oracle_all(oracle, q, targets, sc)
        got = db.scores(q, s)
        bad = [(i, int(got[i]), exp[i]["score"], len(targets[i])) for i in range(len(targets)) if int(got[i]) != exp[i]["score"]]
        assert not bad, bad[:10]
        hits = db.search_topk(q, s, k)
        order = _topk(exp, k)
        assert [h["target_index"] for h in hits] == order
        for h, i in zip(hits, order):
            e = exp[i]
            assert (h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"], h["length"]) == (
                e["score"], e["max_x"], e["max_y"], e["q"], e["t"], len(e["q"]))
    finally:
        db.close()


def test_small_db_single_pass(oracle):
    rng = random.Random(1)
    q = _rnd(rng, 300)
    targets = [_rnd(rng, rng.randint(1, 250)) for _ in range(300)]
    targets[17] = targets[17][:40] + q[100:220] + targets[17][40:60]
    _run(oracle, q, targets, (-1, 2, -1))


def test_multi_pass_and_length_skew(oracle):
    rng = random.Random(2)
    q = _rnd(rng, 520)
    targets = [_rnd(rng, rng.randint(200, 1400)) for _ in range(200)]
    for i in (3, 50, 77):
        p = rng.randint(0, 100)
        targets[i] = targets[i][:p] + q[rng.randint(0, 100):rng.randint(300, 520)] + targets[i][p:]
    _run(oracle, q, targets, (-1, 2, -1))
    _run(oracle, q, targets, (-3, 5, -2))


def test_quirk_lengths_and_ring_targets(oracle):
    rng = random.Random(3)
    q = _rnd(rng, 100)
    # T % 64 == 0 on the full path (T < 2Q), ring-path targets (T >= 2Q), empty and 1-residue targets
    lens = [64, 128, 192, 63, 65, 127, 129, 199, 200, 201, 256, 320, 1, 0, 2, 500, 1000] * 8
    targets = [_rnd(rng, n) for n in lens]
    for i in range(0, len(targets), 5):
        n = len(targets[i])
        if n >= 20:
            targets[i] = targets[i][:n - 15] + q[40:55]      # a hit that ends on the last column
    _run(oracle, q, targets, (-1, 2, -1))
    # short query: ring path with T < Qp (quirk Q2) and first-row-only hits
    q2 = _rnd(rng, 20)
    targets2 = [_rnd(rng, rng.randint(40, 63)) for _ in range(150)] + [q2[:1] * 64, "C" * 63 + q2[0]]
    _run(oracle, q2, targets2, (-1, 2, -1))
    q3 = "A" + _rnd(rng, 30, "CG")
    targets3 = ["C" * 63 + "A", "G" * 127 + "A", "C" * 64, "A" * 64] + [_rnd(rng, 64, "CGT") for _ in range(40)]
    _run(oracle, q3, targets3, (-1, 2, -1))


def test_non_acgt_sequences_take_the_exact_path(oracle):
    rng = random.Random(4)
    q = _rnd(rng, 150)
    targets = [_rnd(rng, rng.randint(50, 300)) for _ in range(140)]
    targets[5] = targets[5][:20] + "NNNN" + q[10:90] + "acgt"
    targets[9] = "xyz" * 30
    _run(oracle, q, targets, (-1, 2, -1))
    qn = q[:70] + "N" + q[70:]
    _run(oracle, qn, targets, (-1, 2, -1))


def test_int16_saturation_reruns_in_int32(oracle):
    rng = random.Random(5)
    core = _rnd(rng, 7000)
    q = core
    targets = [_rnd(rng, 300) for _ in range(130)]
    targets[11] = _rnd(rng, 10) + core[100:6900] + _rnd(rng, 10)   # score 5 * 6800 = 34000 > 32767
    import dlsa_amd
    from dlsa_amd import bridge
    s = dlsa_amd.AlignmentScore(5, -4, -4)
    db = bridge.Database(seqs=targets)
    try:
        got = db.scores(q, s)
        exp = oracle.clean_score_plain(q, targets[11], -4, 5, -4)   # int32 model: the i16 reference wraps here
        assert exp >= 34000 and int(got[11]) == exp
        for i in range(len(targets)):
            if i != 11:
                assert int(got[i]) == oracle.clean_score_plain(q, targets[i], -4, 5, -4)
        hits = db.search_topk(q, s, 1)
        assert hits[0]["target_index"] == 11 and hits[0]["score"] == exp and hits[0]["length"] >= 6800
    finally:
        db.close()


def test_c1_shape_1kb_query_vs_1000_targets(oracle):
    # BASELINE.json configs[0]: 1 query (1 kb) vs 1000 targets (~1 kb), match 2 / mismatch 1 / gap 1, top-k 5
    rng = random.Random(6)
    q = _rnd(rng, 1000)
    targets = [_rnd(rng, rng.randint(900, 1100)) for _ in range(1000)]
    targets[123] = targets[123][:300] + q[200:700] + targets[123][300:500]
    _run(oracle, q, targets, (-1, 2, -1), k=5)


def _expected_hits(oracle, q, targets, sc, k):
    exp = _oracle_all(oracle, q, targets, sc)
    return [(i, exp[i]["score"], exp[i]["max_x"], exp[i]["max_y"], exp[i]["q"], exp[i]["t"]) for i in _topk(exp, k)]


def _as_tuples(hits):
    return [(h["target_index"], h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"]) for h in hits]


def test_batch_of_queries_matches_single_searches(oracle):
    """sw_db_search_topk_batch (a work package: many queries, one database) == one sw_db_search_topk per query ==
    the oracle; queries of different lengths, one of them outside the packed fast path (non-ACGT)."""
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(5)
    sc = (-1, 2, -1)
    s = dlsa_amd.AlignmentScore(sc[1], sc[2], sc[0])
    targets = [_rnd(rng, rng.randint(30, 400)) for _ in range(400)]
    queries = [_rnd(rng, n) for n in (150, 37, 260, 64)] + ["ACGTNNACGTACGGT" * 4, ""]
    targets[11] = targets[11][:20] + queries[0][10:140] + targets[11][20:]
    targets[12] = queries[2][100:230]
    db = bridge.Database(seqs=targets)
    try:
        got = db.search_topk_batch(queries, s, 4)
        assert len(got) == len(queries)
        for q, hits in zip(queries, got):
            assert _as_tuples(hits) == _expected_hits(oracle, q, targets, sc, 4), q[:20]
            assert _as_tuples(hits) == _as_tuples(db.search_topk(q, s, 4))
        assert db.search_topk_batch([], s, 4) == []
    finally:
        db.close()


def test_device_topk_ties_reruns_and_k_larger_than_hits(oracle):
    """The on-device selection (histogram threshold + candidate/tie lists) must give exactly the order the host
    rule defines: score descending, then database index ascending -- with many equal scores, with re-run targets
    (saturated, quirk lengths, non-ACGT) competing for the top, and with k above the number of targets."""
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(6)
    sc = (-1, 2, -1)
    s = dlsa_amd.AlignmentScore(sc[1], sc[2], sc[0])
    q = _rnd(rng, 200)
    base = [_rnd(rng, 90) for _ in range(6)]
    targets = [base[i % 6] for i in range(600)]            # 100 copies of 6 sequences: ties everywhere
    targets[301] = q[50:150]                                 # a clear winner
    targets[302] = "N" + q[40:160]                           # non-ACGT -> exact path, second
    targets[5] = _rnd(rng, 64 * 3)                           # T % 64 == 0 on the full path (quirk)
    targets[77] = _rnd(rng, 450)                             # ring path (T >= 2Q)
    db = bridge.Database(seqs=targets)
    try:
        for k in (1, 5, 40, 700):
            assert _as_tuples(db.search_topk(q, s, k)) == _expected_hits(oracle, q, targets, sc, k), k
    finally:
        db.close()


def test_device_topk_tie_list_overflow_falls_back(oracle):
    """More than 65536 targets with the threshold score: the tie list overflows and the library must fall back to
    the full download -- same answer (the smallest database indices among the ties)."""
    import dlsa_amd
    from dlsa_amd import bridge
    sc = (-1, 2, -1)
    s = dlsa_amd.AlignmentScore(sc[1], sc[2], sc[0])
    q = "ACGTTGCAAGGCTTAACCGG"
    n = 70000
    targets = ["TTGCAAGG"] * n
    targets[12345] = "ACGTTGCAAGGCTT"
    db = bridge.Database(seqs=targets)
    try:
        hits = db.search_topk(q, s, 4)
        assert [h["target_index"] for h in hits] == [12345, 0, 1, 2]
        e = oracle.align("lowmem", q, targets[0], *sc)
        assert (hits[1]["score"], hits[1]["query_aln"], hits[1]["target_aln"]) == (e["score"], e["q"], e["t"])
    finally:
        db.close()


@pytest.mark.parametrize("sc", [(-2, 1, -1), (-1, 1, -3), (-4, 3, -1), (-1, 7, -2), (0, 2, -1), (-1, 2, 0)])
def test_other_scorings_like_the_worker_fallback_chain(oracle, sc):
    """Scorings besides match 2 / mismatch 1 / gap 1: the capacity benchmark's (worker_rust_benchmark.go:10-18), steep
    and flat penalties, and two outside the packed kernel's domain (gap 0: the ring path panics in the reference and
    the worker falls back to find_alignment_simd; mismatch 0).  Expected = the oracle's fallback chain per pair."""
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(hash(sc) & 0xFFFF)
    gap, match, miss = sc
    q = _rnd(rng, 150)
    targets = [_rnd(rng, rng.choice([rng.randint(1, 280), 64, 128, 320, 640])) for _ in range(260)]
    targets[3] = targets[3][:15] + q[20:120] + targets[3][15:40]
    targets[200] = q[:60] + "T" + q[60:]
    s = dlsa_amd.AlignmentScore(match, miss, gap)
    exp = [oracle.align_with_fallback(q, t, gap, match, miss) for t in targets]
    assert all(e is not None for e in exp)
    db = bridge.Database(seqs=targets)
    try:
        got = db.scores(q, s)
        bad = [(i, int(got[i]), exp[i]["score"], len(targets[i])) for i in range(len(targets)) if int(got[i]) != exp[i]["score"]]
        assert not bad, bad[:10]
        hits = db.search_topk(q, s, 6)
        order = sorted(range(len(exp)), key=lambda i: (-exp[i]["score"], i))[:6]
        assert [h["target_index"] for h in hits] == order
        for h, i in zip(hits, order):
            e = exp[i]
            assert (h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"]) == (
                e["score"], e["max_x"], e["max_y"], e["q"], e["t"]), (i, len(targets[i]))
    finally:
        db.close()


def test_chained_units_in_every_scheduling_regime(oracle, monkeypatch):
    """A database whose groups span several 16-strip units (targets of ~1 kb) with 16-wave workgroups, under every
    ticket order the library can choose: one lane (each unit trails its producer with an adjacent ticket), a few
    lanes, and the default (one lane per group, the producer has finished).  The scores must not depend on it."""
    import numpy as np
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(11)
    nrng = np.random.default_rng(11)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    q = _rnd(rng, 700)
    n = 6000                                   # 47 groups x ~63 strips > 2048 strips: 16-wave workgroups
    lens = nrng.integers(900, 1101, size=n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    flat = lut[nrng.integers(0, 4, size=int(offs[-1]), dtype=np.uint8)].copy()
    qb = np.frombuffer(q.encode(), dtype=np.uint8)
    for i in (7, 3000, 5999):                  # homologs that cross unit boundaries (columns 200..800)
        o = int(offs[i]) + 200
        flat[o:o + 600] = qb[50:650]
    tgt = lambda i: flat[int(offs[i]):int(offs[i + 1])].tobytes().decode()  # noqa: E731
    sc = (-1, 2, -1)
    s = dlsa_amd.AlignmentScore(sc[1], sc[2], sc[0])
    db = bridge.Database(flat=flat, offsets=offs)
    try:
        ref = None
        for lanes in ("1", "3", "40", None):
            if lanes is None:
                monkeypatch.delenv("SW_HIP_SCORE_LANES", raising=False)
            else:
                monkeypatch.setenv("SW_HIP_SCORE_LANES", lanes)
            got = np.array(db.scores(q, s))
            if ref is None:
                ref = got
                for i in [7, 3000, 5999] + random.Random(12).sample(range(n), 60):
                    assert int(got[i]) == oracle.align("lowmem", q, tgt(i), *sc)["score"], i
            else:
                assert np.array_equal(ref, got), lanes
            hits = db.search_topk(q, s, 3)
            assert sorted(h["target_index"] for h in hits) == [7, 3000, 5999]
    finally:
        db.close()


def test_query_batches_share_one_score_pass(oracle):
    """Consecutive short queries of a batch run as ONE launch of the score kernel (stacked row blocks, one score array
    per query: sw_score.hip MULTI, db_topk_phase): same hits as one search per query and as the oracle -- queries of
    equal and of different lengths, blocks that end inside a 12-row block, units chained through HBM, targets whose
    length is a multiple of 64 (the full-matrix quirk column depends on the query length, so such a length between
    2 Qmin and 2 Qmax splits a batch), an int16 saturation inside a batch."""
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(77)
    targets = [_rnd(rng, rng.randint(20, 700)) for _ in range(560)] + [_rnd(rng, n) for n in (64, 128, 192, 256, 320, 640) for _ in range(4)]
    q150 = [_rnd(rng, 150) for _ in range(6)]
    qmix = [_rnd(rng, n) for n in (100, 101, 110, 96, 12, 1)]     # [2 * 1, 2 * 110) holds 64, 128, 192: splits
    qsplit = [_rnd(rng, 127), _rnd(rng, 130)]                     # 256 lies in [254, 260): two passes
    for i, q in enumerate(q150):
        targets[10 + i] = targets[10 + i][:30] + q[5:140] + targets[10 + i][30:50]
    targets[40] = qmix[0]
    targets[41] = qmix[2][3:] + "ACGT"
    queries = q150 + qmix + qsplit
    sc = (-1, 2, -1)
    s = dlsa_amd.AlignmentScore(sc[1], sc[2], sc[0])
    db = bridge.Database(seqs=targets)
    try:
        got = db.search_topk_batch(queries, s, 5)
        launches = bridge.last_stats()["score_launches"]
        assert launches < len(queries), launches          # batches were formed ...
        assert launches >= 3, launches                    # ... and split where the quirk column differs
        for q, hits in zip(queries, got):
            assert _as_tuples(hits) == _expected_hits(oracle, q, targets, sc, 5), len(q)
            assert _as_tuples(hits) == _as_tuples(db.search_topk(q, s, 5)), len(q)
        # the same six queries alone: exactly one pass
        got6 = db.search_topk_batch(q150, s, 5)
        assert bridge.last_stats()["score_launches"] == 1
        assert [_as_tuples(h) for h in got6] == [_as_tuples(h) for h in got[:6]]
    finally:
        db.close()
    # saturation inside a batch: 900 matches of 40 pass 32767, the pair is re-run in int32
    sc2 = (-30, 40, -30)
    s2 = dlsa_amd.AlignmentScore(sc2[1], sc2[2], sc2[0])
    qa, qb = _rnd(rng, 900), _rnd(rng, 900)
    t2 = [_rnd(rng, rng.randint(100, 1200)) for _ in range(130)]
    t2[5] = "ACG" + qa + "TT"
    db = bridge.Database(seqs=t2)
    try:
        got = db.search_topk_batch([qa, qb], s2, 3)
        assert bridge.last_stats()["score_launches"] == 1
        exp = oracle.clean_score_plain(qa, t2[5], -30, 40, -30)   # int32 model: the i16 reference wraps here
        assert exp == 36000 and got[0][0]["target_index"] == 5 and got[0][0]["score"] == exp and got[0][0]["length"] == 900
        assert _as_tuples(got[1]) == _expected_hits(oracle, qb, t2, sc2, 3)
        for q, hits in zip([qa, qb], got):
            assert _as_tuples(hits) == _as_tuples(db.search_topk(q, s2, 3))
    finally:
        db.close()
