"""BASELINE.json's configurations at (or near) full size on the GPU, checked through what does not need a CPU
pass over 10^12 cells: planted hits that must come out on top in a known order, alignments that re-derive their
own score and end where max_x / max_y say (master/utils/verify.py:24-48 restated), the on-device top-k against
the full score vector, and the oracle on a seeded SAMPLE of the pairs (bit-exact score, coordinates, strings).

C2: 1 x 10 kb vs 100 000 x 1 kb (the bench workload, full size).   C3: 32 queries vs 10^6 targets of 200-5000 nt
(full size on one GPU).   C4: one 100 kb x 100 kb pair.   C5: 150 nt queries vs 150 nt reads, scaled to
100 queries x 10^6 reads (the full 1000 x 10^7 is the same code 100 x longer)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SC = (-1, 2, -1)  # gap, match, miss: match 2 / mismatch 1 / gap 1
LUT = np.frombuffer(b"ATCG", dtype=np.uint8)


def _synth(rng, lens):
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    return LUT[rng.integers(0, 4, size=int(offs[-1]), dtype=np.uint8)].copy(), offs


def _mutate(rng, seq, rate):
    seq = seq.copy()
    hit = rng.random(len(seq)) < rate
    seq[hit] = LUT[rng.integers(0, 4, size=int(hit.sum()))]
    return seq


def _verify(q, t, score, max_x, max_y, qa, ta):
    gap, match, miss = SC
    s = 0
    for a, b in zip(qa, ta):
        s += gap if (a == "-" or b == "-") else (match if a == b else miss)
    assert s == score
    qs, ts = qa.replace("-", ""), ta.replace("-", "")
    assert q[max_x + 1 - len(qs):max_x + 1] == qs
    assert t[max_y + 1 - len(ts):max_y + 1] == ts


def _score():
    import dlsa_amd
    return dlsa_amd.AlignmentScore(SC[1], SC[2], SC[0])


def test_c2_full_size_planted_hits_sample_parity_and_topk(oracle):
    from dlsa_amd import bridge
    rng = np.random.default_rng(0xC2)
    n, T, Q = 100_000, 1000, 10_000
    query = _synth(rng, np.array([Q]))[0]
    flat, offs = _synth(rng, np.full(n, T))
    planted = {}
    for rank, (idx, length) in enumerate([(77_777, 900), (5, 800), (99_999, 700), (31_337, 600), (50_000, 500), (12, 400)]):
        start = 500 + 1000 * rank
        frag = _mutate(rng, query[start:start + length], 0.04)
        flat[idx * T + 50: idx * T + 50 + length] = frag
        planted[idx] = length
    q = query.tobytes().decode()
    tgt = lambda i: flat[i * T:(i + 1) * T].tobytes().decode()  # noqa: E731
    db = bridge.Database(flat=flat, offsets=offs)
    try:
        hits = db.search_topk(q, _score(), 5)
        assert [h["target_index"] for h in hits] == [77_777, 5, 99_999, 31_337, 50_000]
        for h in hits:
            e = oracle.clean_align(0, q, tgt(h["target_index"]), *SC)
            assert (h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"]) == (
                e["score"], e["max_x"], e["max_y"], e["q"], e["t"])
            _verify(q, tgt(h["target_index"]), h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"])
        scores = db.scores(q, _score())                       # full download, host classification
        order = sorted(range(n), key=lambda i: (-int(scores[i]), i))[:5]
        assert order == [h["target_index"] for h in hits]     # == the on-device selection
        sample = random.Random(2).sample(range(n), 40) + [12]
        for i in sample:
            assert int(scores[i]) == oracle.clean_align(0, q, tgt(i), *SC)["score"], i
    finally:
        db.close()


def test_c3_32_queries_vs_a_million_length_skewed_targets(oracle):
    from dlsa_amd import bridge
    rng = np.random.default_rng(0xC3)
    n, nq, Q = 1_000_000, 32, 1000
    lens = rng.integers(200, 5001, size=n)
    flat, offs = _synth(rng, lens)
    queries = [_synth(rng, np.array([Q]))[0] for _ in range(nq)]
    where = {}
    for qi in range(nq):  # query qi has one strong and one weaker homolog somewhere in the database
        for k, (length, rate) in enumerate([(880, 0.02), (650, 0.03)]):  # unrelated 5 kb targets reach ~900
            while True:
                idx = int(rng.integers(0, n))
                if lens[idx] >= length + 20 and idx not in where:
                    break
            where[idx] = qi
            o = int(offs[idx]) + 10
            flat[o:o + length] = _mutate(rng, queries[qi][60:60 + length], rate)
            where.setdefault(("q", qi), []).append(idx)
    qs = [x.tobytes().decode() for x in queries]
    tgt = lambda i: flat[int(offs[i]):int(offs[i + 1])].tobytes().decode()  # noqa: E731
    db = bridge.Database(flat=flat, offsets=offs)
    try:
        res = db.search_topk_batch(qs, _score(), 5)
        assert len(res) == nq
        for qi, hits in enumerate(res):
            assert [h["target_index"] for h in hits[:2]] == where[("q", qi)], qi
            assert all(hits[i]["score"] >= hits[i + 1]["score"] for i in range(4))
            for h in hits[:2]:
                _verify(qs[qi], tgt(h["target_index"]), h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"])
        for qi in (0, 17, 31):  # the oracle on the reported hits of three queries: bit-exact
            for h in res[qi]:
                e = oracle.clean_align(0, qs[qi], tgt(h["target_index"]), *SC)
                assert (h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"]) == (
                    e["score"], e["max_x"], e["max_y"], e["q"], e["t"])
        scores = db.scores(qs[3], _score())  # one full score vector: ring-path and full-path targets, quirk lengths
        pick = random.Random(3).sample(range(n), 25) + [int(i) for i in np.nonzero(lens % 64 == 0)[0][:5]]
        for i in pick:
            assert int(scores[i]) == oracle.clean_align(0, qs[3], tgt(i), *SC)["score"], (i, int(lens[i]))
        order = sorted(range(n), key=lambda i: (-int(scores[i]), i))[:5]
        assert order == [h["target_index"] for h in res[3]]
    finally:
        db.close()


def test_c4_single_long_pair_100kb(oracle):
    import dlsa_amd
    rng = np.random.default_rng(0xC4)
    sc = _score()
    # (a) 100 kb x 100 kb, a 5 % mutated copy: the score leaves int16, int32 path; self-consistency at full size
    q = _synth(rng, np.array([100_000]))[0]
    t = _mutate(rng, q, 0.05)
    qs, ts = q.tobytes().decode(), t.tobytes().decode()
    res, err = dlsa_amd.find_rust_alignment_simd_low_mem(qs, ts, sc)
    assert err is None, err
    assert res.score32 > 32767 and res.score == res.score32 & 0xFFFF
    _verify(qs, ts, res.score32, res.max_x, res.max_y, res.query, res.target)
    # (b) 100 kb x 100 kb of unrelated sequences: long gappy alignment, self-consistent
    t2 = _synth(rng, np.array([100_000]))[0].tobytes().decode()
    res2, err = dlsa_amd.find_rust_alignment_simd_low_mem(qs, t2, sc)
    assert err is None, err
    _verify(qs, t2, res2.score32, res2.max_x, res2.max_y, res2.query, res2.target)
    # (c) the same kernels against the oracle where a CPU matrix still fits: 30 kb x 30 kb (118 strips)
    qa, ta = qs[:30_000], ts[2_000:32_000]
    e = oracle.clean_align(0, qa, ta, *SC)
    res3, err = dlsa_amd.find_rust_alignment_simd_low_mem(qa, ta, sc)
    assert err is None, err
    assert (res3.score32, res3.max_x, res3.max_y) == (e["score"], e["max_x"], e["max_y"])
    assert (res3.query, res3.target) == (e["q"], e["t"])
    # score of (b) against a linear-memory CPU pass over a 40 kb x 40 kb corner
    res4, err = dlsa_amd.find_rust_alignment_simd_low_mem(qs[:40_000], t2[:39_999], sc)
    assert err is None and res4.score32 == oracle.clean_score_plain(qs[:40_000], t2[:39_999], *SC)


def test_c5_short_queries_vs_a_million_reads(oracle):
    from dlsa_amd import bridge
    rng = np.random.default_rng(0xC5)
    n, nq, L = 1_000_000, 100, 150
    flat, offs = _synth(rng, np.full(n, L))
    queries = [_synth(rng, np.array([L]))[0] for _ in range(nq)]
    home = rng.choice(n, size=nq, replace=False)
    for qi in range(nq):  # read home[qi] is the query itself with three substitutions
        r = queries[qi].copy()
        for p in (10, 70, 140):
            r[p] = LUT[(int(np.nonzero(LUT == r[p])[0][0]) + 1) % 4]
        flat[int(home[qi]) * L:(int(home[qi]) + 1) * L] = r
    qs = [x.tobytes().decode() for x in queries]
    tgt = lambda i: flat[i * L:(i + 1) * L].tobytes().decode()  # noqa: E731
    db = bridge.Database(flat=flat, offsets=offs)
    try:
        res = db.search_topk_batch(qs, _score(), 5)
        for qi, hits in enumerate(res):
            h = hits[0]
            assert h["target_index"] == int(home[qi])
            e = oracle.align("lowmem", qs[qi], tgt(h["target_index"]), *SC)   # the line-by-line restatement
            assert (h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"]) == (
                e["score"], e["max_x"], e["max_y"], e["q"], e["t"])
            assert all(hits[i]["score"] >= hits[i + 1]["score"] for i in range(4))
        scores = db.scores(qs[7], _score())
        for i in random.Random(5).sample(range(n), 150):
            assert int(scores[i]) == oracle.align("lowmem", qs[7], tgt(i), *SC)["score"], i
        order = sorted(range(n), key=lambda i: (-int(scores[i]), i))[:5]
        assert order == [h["target_index"] for h in res[7]]
    finally:
        db.close()


def _top_order(scores, k):
    """(score desc, index asc) over a big numpy score vector without a Python sort of 10^7 items"""
    scores = np.asarray(scores)
    kth = np.partition(scores, len(scores) - k)[len(scores) - k]
    cand = np.flatnonzero(scores >= kth)
    return [int(i) for i in cand[np.lexsort((cand, -scores[cand].astype(np.int64)))][:k]]


def test_c5_real_size_ten_million_reads(oracle):
    """BASELINE.json configs[4] at its real database size: 10^7 reads of 150 nt (78 125 lane groups, 781 250 strips,
    a 40 MB score vector), one query and a batch of 8.  Planted reads must come out on top, reported hits are
    oracle-exact, and the on-device selection equals the host selection over the full score vector."""
    from dlsa_amd import bridge
    n, L, nq = 10_000_000, 150, 8
    flat, offs = bridge.synth(0xC55, n, L, L)
    flat = flat.copy()
    qf, qo = bridge.synth(0xC56, nq, L, L)
    queries = [qf[i * L:(i + 1) * L].copy() for i in range(nq)]
    rng = np.random.default_rng(0xC5)
    home = [int(x) for x in rng.choice(n, size=nq, replace=False)]
    home[0], home[1] = 0, n - 1                         # first and last read of the database
    for qi in range(nq):                                 # read home[qi] = the query with two substitutions
        r = queries[qi].copy()
        for p in (20, 100):
            r[p] = LUT[(int(np.nonzero(LUT == r[p])[0][0]) + 1) % 4]
        flat[home[qi] * L:(home[qi] + 1) * L] = r
    qs = [x.tobytes().decode() for x in queries]
    tgt = lambda i: flat[i * L:(i + 1) * L].tobytes().decode()  # noqa: E731
    db = bridge.Database(flat=flat, offsets=offs)
    try:
        single = db.search_topk(qs[0], _score(), 5)
        res = db.search_topk_batch(qs, _score(), 5)
        assert [h["target_index"] for h in single] == [h["target_index"] for h in res[0]]
        for qi, hits in enumerate(res):
            assert hits[0]["target_index"] == home[qi] and len(hits) == 5
            for h in hits:
                e = oracle.align("lowmem", qs[qi], tgt(h["target_index"]), *SC)
                assert (h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"]) == (
                    e["score"], e["max_x"], e["max_y"], e["q"], e["t"])
        scores = db.scores(qs[5], _score())              # 10^7 scores through the host path
        assert _top_order(scores, 5) == [h["target_index"] for h in res[5]]
        for i in random.Random(6).sample(range(n), 60) + [0, n - 1, home[5]]:
            assert int(scores[i]) == oracle.align("lowmem", qs[5], tgt(i), *SC)["score"], i
    finally:
        db.close()


def test_c3_sharded_path_at_full_size():
    """BASELINE.json configs[2] through the multi-GPU handle: the 10^6-target length-skewed database over three
    (logical) shards, 8 of the 32 queries; must equal the unsharded search hit for hit, shards balanced within 2 %."""
    from dlsa_amd import bridge
    n, nq, Q = 1_000_000, 8, 1000
    flat, offs = bridge.synth(0xC31, n, 200, 5000)
    qf, qo = bridge.synth(0xC32, nq, Q, Q)
    flat = flat.copy()
    rng = np.random.default_rng(0xC3)
    lens = np.diff(offs.astype(np.int64))
    for qi in range(nq):                                  # one homolog per query
        idx = int(rng.choice(np.flatnonzero(lens >= 1000)))
        o = int(offs[idx]) + 5
        flat[o:o + 900] = _mutate(rng, qf[qi * Q + 50:qi * Q + 950], 0.03)
    qs = [qf[i * Q:(i + 1) * Q].tobytes().decode() for i in range(nq)]
    node = bridge.Node(devices=[0, 0, 0])
    ndb = bridge.NodeDatabase(node, flat=flat, offsets=offs)
    try:
        info = ndb.shard_info()
        res = [i["residues"] for i in info]
        assert sum(i["n_targets"] for i in info) == n and (max(res) - min(res)) / (sum(res) / 3) < 0.02
        got = ndb.search_topk_batch(qs, _score(), 5)
    finally:
        ndb.close()
        node.close()
    db = bridge.Database(flat=flat, offsets=offs)
    try:
        ref = db.search_topk_batch(qs, _score(), 5)
    finally:
        db.close()
    key = lambda hits: [(h["target_index"], h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"]) for h in hits]  # noqa: E731
    assert [key(a) for a in got] == [key(b) for b in ref]
    for qi, hits in enumerate(got):
        _verify(qs[qi], flat[int(offs[hits[0]["target_index"]]):int(offs[hits[0]["target_index"] + 1])].tobytes().decode(),
                hits[0]["score"], hits[0]["max_x"], hits[0]["max_y"], hits[0]["query_aln"], hits[0]["target_aln"])
