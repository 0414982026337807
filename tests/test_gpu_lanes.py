"""GPU parity of the one-pair-per-lane exact kernel (dlsa_amd/csrc/sw_lanes.hip) against the oracle, through the C ABI
(sw_align_batch): random reads, the padded-length quirks (T % 64 == 0), ties between maximal cells, pairs the kernel
must hand back to the whole-matrix path, and the automatic choice for a big batch.  SW_HIP_LANES=1 forces the lane
path for every eligible pair (the library reads the variable at every call), 0 forbids it."""
import os
import random

import pytest

pytestmark = pytest.mark.gpu

PATHS = {0: "lowmem", 1: "simd"}


@pytest.fixture
def lanes_env():
    old = os.environ.get("SW_HIP_LANES")
    yield lambda v: os.environ.__setitem__("SW_HIP_LANES", v)
    if old is None:
        os.environ.pop("SW_HIP_LANES", None)
    else:
        os.environ["SW_HIP_LANES"] = old


def _rnd(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def _tuple(r):
    return None if r is None else (r.query, r.target, r.score, r.max_x, r.max_y)


def _expect(oracle, path, q, t, sc):
    e = oracle.align(PATHS[path], q, t, *sc)
    return None if e is None else (e["q"], e["t"], e["score"] & 0xFFFF, e["max_x"], e["max_y"])


def _run(qs, ts, sc, path):
    import dlsa_amd
    from dlsa_amd import bridge
    gap, match, miss = sc
    before = bridge.last_stats()["lanes_pairs"]
    res = bridge.align_batch(qs, ts, dlsa_amd.AlignmentScore(match, miss, gap), path)
    return res, bridge.last_stats()["lanes_pairs"] - before


def _check_all(oracle, qs, ts, sc, path, res):
    for i, (q, t) in enumerate(zip(qs, ts)):
        assert _tuple(res[i]) == _expect(oracle, path, q, t, sc), (i, path, sc, len(q), len(t), q, t)


@pytest.mark.parametrize("sc", [(-1, 2, -1), (-2, 1, -1), (-3, 5, -2), (-1, 1, -1), (-5, 4, -1)])
def test_random_reads_both_entries(oracle, lanes_env, sc):
    lanes_env("1")
    rng = random.Random(101 + sc[1])
    for path in (0, 1):
        qs, ts = [], []
        for _ in range(400):
            n, m = rng.randint(1, 210), rng.randint(1, 210)
            q = _rnd(rng, n)
            if rng.random() < 0.5:   # a mutated copy: long alignments with gaps
                t = "".join(c if rng.random() > 0.12 else rng.choice("ACGT-") for c in q).replace("-", "")[:m] or "A"
            else:
                t = _rnd(rng, m)
            qs.append(q)
            ts.append(t)
        res, n_lane = _run(qs, ts, sc, path)
        assert n_lane > 300, n_lane   # (a few tiny ring-path pairs with T < Qp are not eligible)
        _check_all(oracle, qs, ts, sc, path, res)


def test_padded_length_quirks_and_hand_back(oracle, lanes_env):
    """T % 64 == 0 on the full-matrix path: the last column is skipped except H(1, T) (rust/src/algorithm/mod.rs:98,
    134,178); the lane kernel leaves that one cell out and hands the pair back when it could matter (score <= match)."""
    lanes_env("1")
    rng = random.Random(7)
    sc = (-1, 2, -1)
    qs, ts = [], []
    for T in (64, 128, 192):
        for Q in (33, 40, 64, 65, 100, 128, 129, 190):
            qs.append(_rnd(rng, Q)); ts.append(_rnd(rng, T))
            qs.append("A" * Q); ts.append("C" * (T - 1) + "A")       # only H(1, T) scores
            qs.append("A" + "G" * (Q - 1)); ts.append("C" * (T - 1) + "A")
            qs.append("G" * Q); ts.append("C" * T)                   # no hit at all
            q = _rnd(rng, Q); qs.append(q); ts.append((q * 4)[:T])  # the alignment runs into the skipped column
    for path in (0, 1):
        res, n_lane = _run(qs, ts, sc, path)
        assert n_lane > 0
        _check_all(oracle, qs, ts, sc, path, res)


def test_ties_take_the_reference_first_cell(oracle, lanes_env):
    """Several cells reach the maximum: smallest x + j first, then smallest x (rust/src/lib.rs:122), across columns
    of one strip, across strips and across rows."""
    lanes_env("1")
    rng = random.Random(9)
    sc = (-1, 2, -1)
    qs, ts = [], []
    for _ in range(150):
        frag = _rnd(rng, rng.randint(3, 12))
        n_rep = rng.randint(2, 5)
        fill = lambda k: "".join(rng.choice("NX") for _ in range(k))  # noqa: E731  (never matches a fragment)
        q = fill(rng.randint(0, 30)) + frag + fill(rng.randint(0, 40)) + (frag if rng.random() < 0.5 else "") + fill(rng.randint(0, 9))
        t = fill(rng.randint(0, 20)).lower()
        for _ in range(n_rep):
            t += frag + fill(rng.randint(1, 35)).lower()
        qs.append(q); ts.append(t)
        qs.append(t); ts.append(q)
    for path in (0, 1):
        res, n_lane = _run(qs, ts, sc, path)
        assert n_lane > 0
        _check_all(oracle, qs, ts, sc, path, res)


def test_mixed_batch_with_ineligible_pairs(oracle, lanes_env):
    lanes_env("1")
    rng = random.Random(11)
    sc = (-1, 2, -1)
    qs, ts = [], []
    for i in range(200):
        q, t = _rnd(rng, rng.randint(20, 160)), _rnd(rng, rng.randint(20, 160))
        if i % 7 == 0:
            q = q[:10] + "é" + q[10:]          # not ASCII: whole-matrix path
        if i % 11 == 0:
            t = _rnd(rng, 3000)                 # ring path, long target
        if i % 13 == 0:
            q, t = _rnd(rng, 20), _rnd(rng, 45)  # ring path with T < Qp: two-phase discovery order
        qs.append(q); ts.append(t)
    res, n_lane = _run(qs, ts, sc, 0)
    assert 0 < n_lane < 200
    _check_all(oracle, qs, ts, sc, 0, res)


def test_lane_path_equals_whole_matrix_path_on_1kb_pairs(oracle, lanes_env):
    """Longer pairs (many strips, boundary column through HBM): both paths of the library and a sample of the oracle."""
    rng = random.Random(13)
    sc = (-1, 2, -1)
    qs = [_rnd(rng, rng.randint(700, 1100)) for _ in range(96)]
    ts = [_rnd(rng, rng.randint(700, 1100)) for _ in range(96)]
    for k in range(0, 96, 3):  # related pairs
        ts[k] = "".join(c if rng.random() > 0.1 else rng.choice("ACGT") for c in qs[k])[: len(ts[k])]
    lanes_env("1")
    a, n_lane = _run(qs, ts, sc, 0)
    assert n_lane == 96
    lanes_env("0")
    b, n0 = _run(qs, ts, sc, 0)
    assert n0 == 0
    assert [_tuple(x) for x in a] == [_tuple(x) for x in b]
    for i in range(0, 96, 8):
        assert _tuple(a[i]) == _expect(oracle, 0, qs[i], ts[i], sc)


def test_big_batch_of_reads_takes_the_lane_path_by_itself(oracle, lanes_env):
    lanes_env("auto")
    rng = random.Random(17)
    sc = (-1, 2, -1)
    reads = [_rnd(rng, 150) for _ in range(300)]
    qs = [reads[rng.randrange(300)] for _ in range(20000)]
    ts = [reads[rng.randrange(300)] if rng.random() < 0.9 else _rnd(rng, rng.randint(100, 150)) for _ in range(20000)]
    res, n_lane = _run(qs, ts, sc, 0)
    assert n_lane == 20000
    for i in range(0, 20000, 97):
        assert _tuple(res[i]) == _expect(oracle, 0, qs[i], ts[i], sc), i
    lanes_env("0")
    ref, _ = _run(qs[:4000], ts[:4000], sc, 0)
    assert [_tuple(x) for x in res[:4000]] == [_tuple(x) for x in ref]


def test_execute_work_through_the_lane_kernel_equals_the_go_worker_on_the_oracle(oracle, lanes_env):
    """sw_execute_work (fallback chain, sign flip, int() conversions) with every eligible pair on the lane kernel:
    empty and non-ASCII sequences take the other paths inside the same call."""
    import test_gpu_worker as tw
    from dlsa_amd import worker
    lanes_env("1")
    rng = random.Random(71)
    work = tw._package(rng, 5, 60)
    w = worker.Worker()
    w.execute_work(work, work.queries)
    details = [d for r in w.sent for d in r.alignments]
    assert len(details) == len(work.queries)
    for comb, d in zip(work.queries, details):
        e = tw._expected(oracle, work, comb)
        a = d.alignment
        assert (a.query_alignment, a.target_alignment) == (e["q"], e["t"]), (comb,)
        assert a.score == e["score"] & 0xFFFF
        assert (a.maxX, a.maxY) == (tw._go_int(e["max_x"]), tw._go_int(e["max_y"]))
        assert d.entry == e["which"]


def test_long_pairs_many_strips_identical_and_unrelated(oracle, lanes_env):
    """3 kb pairs: 188 strips per group chained through the tagged boundary column; identical sequences give one
    long diagonal (score 2 x length, the traceback crosses every strip), unrelated ones a gappy path."""
    lanes_env("1")
    rng = random.Random(73)
    sc = (-1, 2, -1)
    a = _rnd(rng, 3000)
    qs = [a, a, _rnd(rng, 2900), a[:2500], _rnd(rng, 1500)]
    ts = [a, _rnd(rng, 3000), a, a[300:], "".join(c if rng.random() > 0.08 else rng.choice("ACGT") for c in a[:1700])]
    res, n_lane = _run(qs, ts, sc, 1)
    assert n_lane == len(qs)
    _check_all(oracle, qs, ts, sc, 1, res)


def test_node_batch_over_logical_shards_with_the_lane_kernel(oracle, lanes_env):
    """sw_node_align_batch cuts the pair list into one part per shard; every part may take the lane path on its
    device's context (two logical shards on device 0 here)."""
    import dlsa_amd
    from dlsa_amd import bridge
    lanes_env("1")
    rng = random.Random(79)
    sc = (-1, 2, -1)
    qs = [_rnd(rng, rng.randint(40, 200)) for _ in range(300)]
    ts = [_rnd(rng, rng.randint(40, 200)) for _ in range(300)]
    node = bridge.Node(devices=[0, 0])
    try:
        res = bridge.align_batch(qs, ts, dlsa_amd.AlignmentScore(2, -1, -1), 0, node=node)
    finally:
        node.close()
    _check_all(oracle, qs, ts, sc, 0, res)


def test_symbol_block_of_the_batch_or_per_slice_staging(oracle, lanes_env):
    """A batch of >= 1024 pairs is decoded into one page-locked block that the lane path uploads as it is; when the
    block holds much more than the eligible pairs need (long sequences of pairs that go elsewhere), every slice
    stages its own symbols instead.  Same results either way."""
    lanes_env("1")
    rng = random.Random(83)
    sc = (-1, 2, -1)
    reads = [_rnd(rng, rng.randint(30, 90)) for _ in range(200)]
    qs = [reads[rng.randrange(200)] for _ in range(1300)]
    ts = [reads[rng.randrange(200)] for _ in range(1300)]
    a, n_a = _run(qs, ts, sc, 0)                      # the block: 200 reads, far less than 1300 pairs' worth
    assert n_a > 1200, n_a                            # (a few ring-path pairs with T < Qp are not eligible)
    long_t = [_rnd(rng, 50000, "ACGTé") for _ in range(6)]   # not ASCII: these pairs take the whole-matrix path
    qs2 = qs + [_rnd(rng, 24) for _ in range(6)]
    ts2 = ts + long_t
    b, n_b = _run(qs2, ts2, sc, 0)                    # the block: 300 000 symbols for pairs that need 160 000
    assert n_b == n_a
    assert [_tuple(x) for x in a] == [_tuple(x) for x in b[:1300]]
    for i in list(range(0, 1300, 41)) + [1300, 1303]:
        assert _tuple(b[i]) == _expect(oracle, 0, qs2[i], ts2[i], sc), i
