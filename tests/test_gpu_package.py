"""The work-package fast path (dlsa_amd/csrc/sw_package.cpp) against the oracle: pairs of the reference's own synthetic
shape (utils/generate_synthetic_data.py: short queries, long targets) through sw_align_batch / sw_execute_work must
give the oracle's find_alignment_low_memory result field by field -- score, end coordinates, both aligned strings --
although the matrix is filled by the packed score kernel and only windows of it are re-entered with direction bits.
Planted homologs make alignments that cross several windows (continuation rounds), duplicated fragments make equal
maxima in different windows (tie-breaking across windows)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SC = (-1, 2, -1)


def _rnd(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def _mut(rng, s, rate):
    return "".join((rng.choice("ACGT") if rng.random() < rate else c) for c in s)


def _check(oracle, res, q, t, sc=SC):
    e = oracle.align("lowmem", q, t, *sc)
    assert res is not None
    assert (res.score, res.max_x, res.max_y, res.query, res.target) == (e["score"], e["max_x"], e["max_y"], e["q"], e["t"])


def test_package_of_long_targets_matches_the_oracle(oracle):
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(41)
    s = dlsa_amd.AlignmentScore(SC[1], SC[2], SC[0])
    queries = [_rnd(rng, n) for n in (200, 517, 1000, 64)]
    targets = [_rnd(rng, rng.randint(3000, 9000)) for _ in range(20)] + [_rnd(rng, 60000), _rnd(rng, 4096), _rnd(rng, 2048)]
    # homologs: long alignments that cross unit boundaries (every 256 target columns) -> continuation rounds
    targets[0] = targets[0][:1000] + _mut(rng, queries[2][50:950], 0.04) + targets[0][1000:]
    targets[1] = targets[1][:250] + queries[1][10:500] + targets[1][250:]
    # the same fragment twice: equal maxima in different windows, the reference takes the first in its order
    frag = queries[0][20:180]
    targets[2] = targets[2][:300] + frag + targets[2][300:2000] + frag + targets[2][2000:]
    # ... and at the very end / very beginning of a target
    targets[3] = targets[3] + queries[0][:150]
    targets[4] = queries[1][100:400] + targets[4]
    qs = [q for q in queries for _ in targets]
    ts = targets * len(queries)
    res = bridge.align_batch(qs, ts, s, 0)
    st = bridge.package_stats()
    assert st["pairs"] == len(qs) and st["queries"] == len(queries) and st["chunk_escapes"] == 0
    assert st["chunks"] > len(targets)                       # the 60 kb target was cut into overlapping chunks
    assert st["rounds"] > len(queries)                       # some traceback continued to the left
    total = sum(len(q) * len(t) for q, t in zip(qs, ts))
    assert st["window_cells"] < 0.25 * total                 # the exact kernel saw a fraction of the matrix
    for r, q, t in zip(res, qs, ts):
        _check(oracle, r, q, t)


def test_package_mixed_with_ineligible_pairs_and_other_scorings(oracle):
    """Eligible and ineligible pairs in one batch (short targets, non-ACGT, T < 2Q): each takes its own path, results
    in input order; steeper scorings; the worker's ExecuteWork entry."""
    import dlsa_amd
    from dlsa_amd import bridge, worker
    rng = random.Random(42)
    q1, q2 = _rnd(rng, 300), _rnd(rng, 900)
    long_t = [_rnd(rng, rng.randint(20000, 40000)) for _ in range(12)]
    long_t[5] = long_t[5][:7000] + _mut(rng, q2[:800], 0.1) + long_t[5][7000:]
    long_t[6] = long_t[6][:255] + q1[40:260] + long_t[6][255:]          # ends right after a unit boundary
    odd = [_rnd(rng, 500), "N" * 10 + _rnd(rng, 5000), _rnd(rng, 1500), ""]
    targets = long_t + odd
    for sc in ((-1, 2, -1), (-3, 5, -2), (-2, 1, -1)):
        s = dlsa_amd.AlignmentScore(sc[1], sc[2], sc[0])
        qs = [q for q in (q1, q2) for _ in targets]
        ts = targets * 2
        res = bridge.align_batch(qs, ts, s, 0)
        st = bridge.package_stats()
        assert st["pairs"] == 2 * len(long_t) and st["chunk_escapes"] == 0 and st["chunks"] > len(long_t), st
        for r, q, t in zip(res, qs, ts):
            _check(oracle, r, q, t, sc)
    # the worker's entry point (sw_execute_work): same path underneath
    seqs = {f"s{i}": t for i, t in enumerate(long_t)}
    seqs["q"] = q1
    pkg = worker.CompleteWorkPackage(id="w", job_id="j", queries=[worker.QueryTargetType("q", f"s{i}") for i in range(len(long_t))],
                                     match_score=2, mismatch_penalty=1, gap_penalty=1, sequences=seqs)
    out = worker.Worker().align(pkg, pkg.queries)
    assert bridge.package_stats()["pairs"] == len(long_t)
    for d, t in zip(out, long_t):
        e = oracle.align("lowmem", q1, t, -1, 2, -1)
        a = d.alignment
        assert (a.query_alignment, a.target_alignment, a.score, a.maxX, a.maxY, a.length) == (
            e["q"], e["t"], e["score"], e["max_x"], e["max_y"], len(e["q"]))


def test_package_beyond_the_ring_budget_is_sliced_not_dropped(oracle):
    """A target set whose boundary ring would not fit twice the scratch budget is processed in slices of its targets
    (sw_package.cpp), not left to the whole-matrix path.  52 M residues against a 480-nt query want ~330 MB of ring;
    with the budget at 160 MiB the set goes through in two slices: still the fast path (chunks > 0, every pair
    answered by it, more score passes than queries), results equal to the unsliced run's and to the oracle's."""
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(47)
    nrng = np.random.default_rng(47)
    s = dlsa_amd.AlignmentScore(SC[1], SC[2], SC[0])
    queries = [_rnd(rng, 300), _rnd(rng, 480)]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    targets = [lut[nrng.integers(0, 4, size=int(n), dtype=np.uint8)].tobytes().decode() for n in nrng.integers(110000, 150000, size=400)]
    targets[3] = targets[3][:700] + _mut(rng, queries[1][20:460], 0.05) + targets[3][700:]
    targets[317] = targets[317][:90000] + queries[0][5:290] + targets[317][90000:]
    qs = [q for q in queries for _ in targets]
    ts = targets * len(queries)
    whole = bridge.align_batch(qs, ts, s, 0)
    st0 = bridge.package_stats()
    assert st0["pairs"] == len(qs) and st0["queries"] == len(queries), st0
    bridge.set_scratch_mb(160)
    try:
        res = bridge.align_batch(qs, ts, s, 0)
        st = bridge.package_stats()
    finally:
        bridge.set_scratch_mb(8192)
    assert st["chunks"] > 0 and st["pairs"] == len(qs), st
    assert st["queries"] > len(queries), st          # more score passes than queries: the set went through in slices
    key = lambda r: (r.score, r.max_x, r.max_y, r.query, r.target)  # noqa: E731
    assert [key(r) for r in res] == [key(r) for r in whole]
    for i in (3, 317, 400 + 3, 400 + 317, 11, 799):
        _check(oracle, res[i], qs[i], ts[i])


def test_many_small_sets_run_side_by_side_on_context_lanes(oracle):
    """What a combined batch of per-pair callers looks like (worker_impl.go:204-238: every goroutine brings a query and
    a target of its own): as many target sets of one target as there are pairs.  The sets run side by side on the
    device, each on a context lane of its own; every pair still gets exactly the oracle's record --
    homologs that cross windows, a duplicated fragment, a pair whose query is another pair's query, a pair with no
    hit."""
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(43)
    s = dlsa_amd.AlignmentScore(SC[1], SC[2], SC[0])
    qs = [_rnd(rng, rng.randint(200, 1000)) for _ in range(13)]
    ts = [_rnd(rng, rng.randint(10000, 60000)) for _ in range(13)]
    ts[0] = ts[0][:5000] + _mut(rng, qs[0], 0.05) + ts[0][5000:]                      # long alignment, several windows
    ts[1] = ts[1][:300] + qs[1][10:150] + ts[1][300:7000] + qs[1][10:150] + ts[1][7000:]  # equal maxima
    ts[2] = qs[2][:180] + ts[2]                                                      # at the very beginning
    qs[3] = qs[4]                                                                    # the same query object twice
    qs[5], ts[5] = "A" * 300, "C" * 12000                                            # no hit at all
    res = bridge.align_batch(qs, ts, s, 0)
    st = bridge.package_stats()
    assert st["pairs"] == 13 and st["chunk_escapes"] == 0, st
    assert st["queries"] in (12, 13), st      # one pass per set and query (the lanes' counters are summed); the
                                              # query used twice is one set of two targets if the caller interned it
    for r, q, t in zip(res, qs, ts):
        _check(oracle, r, q, t)
    # the same pairs one call at a time (one set, one lane) give the same records
    for i in (0, 1, 5, 9):
        one = bridge.align_batch([qs[i]], [ts[i]], s, 0)[0]
        a, b = one, res[i]
        assert (a.score, a.max_x, a.max_y, a.query, a.target) == (b.score, b.max_x, b.max_y, b.query, b.target)
