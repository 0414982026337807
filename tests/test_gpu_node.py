"""GPU parity of the multi-GPU handle (include/sw_hip.h: sw_node_*): ONE process, the database sharded by target
over several device contexts, per-shard top-k, one host merge, exact alignments of the merged hits on the owning
shard.  The GPU box has one MI355X, so the shards are LOGICAL shards on device 0 (the same code path per shard as on
8 GPUs: own database, own submission, own top-k; only the contexts coincide).  Expected = the unsharded search and
the oracle's find_alignment_low_memory restatement, field by field."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rnd(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def _as_tuples(hits):
    return [(h["target_index"], h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"], h["length"]) for h in hits]


def _expected(oracle, q, targets, sc, k):
    exp = [oracle.align("lowmem", q, t, *sc) for t in targets]
    order = sorted(range(len(exp)), key=lambda i: (-exp[i]["score"], i))[:k]
    return [(i, exp[i]["score"], exp[i]["max_x"], exp[i]["max_y"], exp[i]["q"], exp[i]["t"], len(exp[i]["q"])) for i in order]


@pytest.mark.parametrize("n_shards", [1, 2, 3])
def test_sharded_search_equals_unsharded_and_oracle(oracle, n_shards):
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(21)
    sc = (-1, 2, -1)
    s = dlsa_amd.AlignmentScore(sc[1], sc[2], sc[0])
    # length-skewed targets incl. quirk lengths, empty, non-ACGT (exact path) and ties across shards
    targets = [_rnd(rng, rng.choice([rng.randint(1, 600), 64, 128, 320])) for _ in range(700)]
    targets += ["", "N" * 30, "ACGTNACGT" * 9]
    dup = _rnd(rng, 90)
    for i in (5, 300, 301, 650):
        targets[i] = dup                                     # equal scores on different shards: index order decides
    queries = [_rnd(rng, 150), _rnd(rng, 40), dup[10:80], "ACGTNNACGTACGGT" * 4]
    targets[17] = targets[17][:20] + queries[0][10:140] + targets[17][20:40]
    targets[600] = queries[0][30:120]
    node = bridge.Node(devices=[0] * n_shards)
    assert node.n_shards == n_shards
    ndb = bridge.NodeDatabase(node, seqs=targets)
    db = bridge.Database(seqs=targets)
    try:
        info = ndb.shard_info()
        assert len(info) == n_shards and sum(i["n_targets"] for i in info) == len(targets)
        assert sum(i["residues"] for i in info) == sum(len(t) for t in targets)
        if n_shards > 1:
            assert min(i["n_targets"] for i in info) > 0
        for k in (1, 5, 12):
            got = ndb.search_topk_batch(queries, s, k)
            ref = db.search_topk_batch(queries, s, k)
            assert len(got) == len(queries)
            for q, g, r in zip(queries, got, ref):
                assert _as_tuples(g) == _as_tuples(r), (k, q[:12])
                assert _as_tuples(g) == _expected(oracle, q, targets, sc, k), (k, q[:12])
        assert ndb.search_topk_batch([], s, 3) == []
        st = ndb.last_stats(n_shards - 1)
        assert st["score_launches"] >= 1
    finally:
        db.close()
        ndb.close()
        node.close()


def test_sharded_chained_database_and_k_beyond_a_shard(oracle):
    """~1 kb targets (groups that span several work units of the score kernel) over 3 shards; k larger than what one
    shard would report on its own must still merge to the global order."""
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(22)
    nrng = np.random.default_rng(22)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    sc = (-1, 2, -1)
    s = dlsa_amd.AlignmentScore(sc[1], sc[2], sc[0])
    q = _rnd(rng, 400)
    n = 3000
    lens = nrng.integers(200, 1500, size=n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    flat = lut[nrng.integers(0, 4, size=int(offs[-1]), dtype=np.uint8)].copy()
    qb = np.frombuffer(q.encode(), dtype=np.uint8)
    planted = [3, 1500, 1501, 2999, 77, 78, 79]
    for j, i in enumerate(planted):
        o = int(offs[i]) + 20
        flat[o:o + 150 + 10 * j] = qb[30:180 + 10 * j]
    node = bridge.Node(devices=[0, 0, 0])
    ndb = bridge.NodeDatabase(node, flat=flat, offsets=offs)
    db = bridge.Database(flat=flat, offsets=offs)
    try:
        so = bridge.shard_plan(offs, 3)
        info = ndb.shard_info()
        assert [i["n_targets"] for i in info] == list(np.bincount(so, minlength=3))
        got = ndb.search_topk_batch([q], s, 7)[0]
        ref = db.search_topk_batch([q], s, 7)[0]
        assert _as_tuples(got) == _as_tuples(ref)
        assert sorted(h["target_index"] for h in got) == sorted(planted)
        tgt = lambda i: flat[int(offs[i]):int(offs[i + 1])].tobytes().decode()  # noqa: E731
        for h in got:
            e = oracle.align("lowmem", q, tgt(h["target_index"]), *sc)
            assert (h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"]) == (
                e["score"], e["max_x"], e["max_y"], e["q"], e["t"])
    finally:
        db.close()
        ndb.close()
        node.close()


def test_node_capacity_benchmark_is_integer_valued():
    """The figure a one-process-per-node worker registers (master/api_models/worker.py:7-16 wants an int)."""
    from dlsa_amd import bridge
    node = bridge.Node(devices=[0])
    try:
        cups = node.benchmark_cups()
        assert cups > 1e11 and cups == float(int(cups))
    finally:
        node.close()


@pytest.mark.parametrize("n_shards", [2, 3])
def test_node_batch_and_execute_work_equal_the_single_gpu_calls(oracle, n_shards):
    """sw_node_align_batch / sw_node_execute_work: the pairs of one package are cut over the node's shards (by query,
    balanced by cells), every part is an ordinary batch, and each pair's result is the one the single-GPU call gives
    -- and the oracle's.  Mixed package: fast-path pairs (long targets), short pairs, failures, a non-ASCII pair."""
    import dlsa_amd
    from dlsa_amd import bridge, worker
    rng = random.Random(77)
    sc = (-1, 2, -1)
    s = dlsa_amd.AlignmentScore(sc[1], sc[2], sc[0])
    queries = [_rnd(rng, n) for n in (120, 300, 64, 900)]
    long_t = [_rnd(rng, rng.randint(9000, 22000)) for _ in range(6)]
    long_t[2] = long_t[2][:5000] + queries[3][100:800] + long_t[2][5000:]
    short_t = [_rnd(rng, rng.randint(1, 700)) for _ in range(40)] + ["", "N" * 12, "ACGTÄCGT" * 5]
    targets = long_t + short_t
    qs = [q for q in queries for _ in targets]
    ts = targets * len(queries)
    node = bridge.Node(devices=[0] * n_shards)
    try:
        got = bridge.align_batch(qs, ts, s, 0, node=node)
        st = bridge.package_stats()
        assert st["pairs"] == len(queries) * len(long_t)          # summed over the shards' threads
        ref = bridge.align_batch(qs, ts, s, 0)
        for g, r, q, t in zip(got, ref, qs, ts):
            assert (g is None) == (r is None)
            if g is None:
                continue
            assert (g.score, g.max_x, g.max_y, g.query, g.target) == (r.score, r.max_x, r.max_y, r.query, r.target)
            e = oracle.align("lowmem", q, t, *sc)
            assert (g.score, g.max_x, g.max_y, g.query, g.target) == (e["score"], e["max_x"], e["max_y"], e["q"], e["t"])
        # the worker's entry: one package, every pair
        seqs = {f"s{i}": v for i, v in enumerate(queries + targets)}
        pairs = [worker.QueryTargetType(f"s{qi}", f"s{len(queries) + ti}")
                 for qi in range(len(queries)) for ti in range(len(targets))]
        work = worker.CompleteWorkPackage("wp", "job", pairs, 2, 1, 1, seqs)
        a = worker.Worker(node=node).align(work, pairs)
        b = worker.Worker().align(work, pairs)
        assert len(a) == len(pairs)
        for x, y in zip(a, b):
            assert x.combination == y.combination and x.entry == y.entry and x.score32 == y.score32
            assert vars(x.alignment) == vars(y.alignment)
    finally:
        node.close()
