"""The packed database on disk (include/sw_hip.h: sw_db_save / sw_db_open, sw_node_db_save / sw_node_db_open;
SURVEY.md 8f rank 2): a saved and re-opened database must give identical scores, top-k and alignments -- for
packed sequences, for the escape list (non-ACGT, empty) and per shard -- and a damaged file must be refused with
an error, not uploaded."""
import os
import random
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rnd(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def _as_tuples(hits):
    return [(h["target_index"], h["score"], h["max_x"], h["max_y"], h["query_aln"], h["target_aln"], h["length"]) for h in hits]


def test_saved_database_round_trips(tmp_path, oracle):
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(31)
    s = dlsa_amd.AlignmentScore(2, -1, -1)
    targets = [_rnd(rng, rng.choice([rng.randint(1, 700), 64, 128, 1500])) for _ in range(900)]
    targets += ["", "NNNNACGT" * 5, "acgtacgt", "ACGTüACGT"]
    q = _rnd(rng, 220)
    targets[40] = targets[40][:10] + q[20:200] + targets[40][10:30]
    targets[901] = "N" + q[50:150]
    path = str(tmp_path / "db.swdb")
    db = bridge.Database(seqs=targets)
    try:
        db.save(path)
        fasta_bytes = sum(len(t.encode()) for t in targets)
        assert os.path.getsize(path) < 0.45 * fasta_bytes          # 2 bits per residue + tables
        db2 = bridge.Database(path=path)
        try:
            assert db2.n == len(targets)
            assert np.array_equal(db.scores(q, s), db2.scores(q, s))
            for k in (1, 6):
                assert _as_tuples(db.search_topk(q, s, k)) == _as_tuples(db2.search_topk(q, s, k))
            h = db2.search_topk(q, s, 2)
            for hit in h:   # the host copy of the sequences was rebuilt from the strip stream: alignments still exact
                e = oracle.align("lowmem", q, targets[hit["target_index"]], -1, 2, -1)
                assert (hit["score"], hit["query_aln"], hit["target_aln"]) == (e["score"], e["q"], e["t"])
            # a second generation (save of a re-opened database) is byte-identical
            path2 = str(tmp_path / "db2.swdb")
            db2.save(path2)
            assert open(path, "rb").read() == open(path2, "rb").read()
        finally:
            db2.close()
    finally:
        db.close()
    # damaged files are refused
    raw = bytearray(open(path, "rb").read())
    for name, mutate in (("magic", lambda b: b.__setitem__(0, 0x58)),
                         ("truncated", lambda b: b.__delitem__(slice(len(b) // 2, len(b)))),
                         ("length", lambda b: b.__setitem__(slice(80, 84), (10 ** 9).to_bytes(4, "little")))):
        b = bytearray(raw)
        mutate(b)
        p = str(tmp_path / f"bad_{name}.swdb")
        open(p, "wb").write(bytes(b))
        with pytest.raises(bridge.SwHipError):
            bridge.Database(path=p)
    with pytest.raises(bridge.SwHipError):
        bridge.Database(path=str(tmp_path / "missing.swdb"))


def test_shard_files_round_trip(tmp_path):
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(32)
    s = dlsa_amd.AlignmentScore(2, -1, -1)
    targets = [_rnd(rng, rng.randint(30, 900)) for _ in range(1200)] + ["ACGTNACGT"]
    queries = [_rnd(rng, 180), _rnd(rng, 60)]
    targets[100] = queries[0][10:170]
    node = bridge.Node(devices=[0, 0, 0])
    ndb = bridge.NodeDatabase(node, seqs=targets)
    d = str(tmp_path / "node_db")
    try:
        ndb.save(d)
        assert sorted(os.listdir(d)) == ["node.swnode", "shard-000.swdb", "shard-001.swdb", "shard-002.swdb"]
        t0 = time.perf_counter()
        ndb2 = bridge.NodeDatabase(node, path=d)
        load_s = time.perf_counter() - t0
        try:
            assert ndb2.n == len(targets) and ndb2.shard_info() == ndb.shard_info()
            a, b = ndb.search_topk_batch(queries, s, 5), ndb2.search_topk_batch(queries, s, 5)
            assert [_as_tuples(x) for x in a] == [_as_tuples(x) for x in b]
            assert load_s < 30
        finally:
            ndb2.close()
        node2 = bridge.Node(devices=[0, 0])
        try:
            with pytest.raises(bridge.SwHipError, match="different number of shards"):
                bridge.NodeDatabase(node2, path=d)
        finally:
            node2.close()
    finally:
        ndb.close()
        node.close()
