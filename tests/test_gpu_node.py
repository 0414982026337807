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
