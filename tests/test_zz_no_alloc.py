"""The allocation-free entry point sw_db_scores_noalloc (include/sw_hip.h) against the entry point it is a variant
of, sw_db_scores, through the C ABI: bit-exact scores on two calls in a row, the second with a smaller input; the
fixed scratch keeps the first call's contents, so the second call proves that everything that must start at zero is
cleared.  Covers chained work units (targets longer than one unit), int32 re-runs (saturation, padded-length
quirks) and the stated limits."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rnd(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def test_noalloc_scores_equal_sw_db_scores_on_two_calls_in_a_row(oracle):
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(77)
    s = dlsa_amd.AlignmentScore(2, -1, -1)
    # call 1: the larger input -- 2000 targets of ~1 kb (groups that span several units: the boundary ring is used),
    # quirk lengths (T % 64 == 0 on the full path), an empty target, a homolog that saturates int16 with this scoring
    q1 = _rnd(rng, 900)
    big = [_rnd(rng, rng.choice([rng.randint(700, 1300), 960, 1024, 64, 128])) for _ in range(2000)]
    big[10] = ""
    big[500] = big[500][:100] + q1[50:850] + big[500][100:200]
    s_sat = dlsa_amd.AlignmentScore(50, -40, -40)                     # 800 matches x 50 = 40 000 > 32767
    # call 2: the smaller input -- fewer, shorter targets and a shorter query
    q2 = _rnd(rng, 130)
    small = [_rnd(rng, rng.randint(1, 300)) for _ in range(300)] + [_rnd(rng, 64), _rnd(rng, 192)]
    small[7] = q2[20:110]
    db1 = bridge.Database(seqs=big)
    db2 = bridge.Database(seqs=small)
    try:
        assert bridge.lib().sw_hip_noalloc_scratch_bytes() <= 128 << 20        # the stated, modest maximum
        for db, q, sc, seqs in ((db1, q1, s, big), (db2, q2, s, small), (db1, q1, s_sat, big), (db2, q2, s_sat, small)):
            ref = db.scores(q, sc)
            got = db.scores_noalloc(q, sc)
            assert np.array_equal(ref, got), np.flatnonzero(ref != got)[:10]
        # ... and against the oracle on a sample (sw_db_scores itself is checked in test_gpu_db_parity.py)
        got = db2.scores_noalloc(q2, s)
        for i in [7, 300, 301] + random.Random(1).sample(range(300), 25):
            assert int(got[i]) == oracle.align("lowmem", q2, small[i], -1, 2, -1)["score"], i
        st = bridge.last_stats()
        assert st["score_launches"] == 1
        # a short query against short targets: the ring path with T < Qp needs the exact kernel for EVERY target
        q3 = _rnd(rng, 20)
        tiny = [_rnd(rng, rng.randint(40, 63)) for _ in range(40)]
        db3 = bridge.Database(seqs=tiny)
        try:
            assert np.array_equal(db3.scores(q3, s), db3.scores_noalloc(q3, s))
            assert bridge.last_stats()["rerun_pairs"] == 40
        finally:
            db3.close()
    finally:
        db1.close()
        db2.close()


def test_noalloc_limits_are_errors_not_allocations():
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(78)
    s = dlsa_amd.AlignmentScore(2, -1, -1)
    db = bridge.Database(seqs=[_rnd(rng, 400) for _ in range(200)])
    dbn = bridge.Database(seqs=[_rnd(rng, 100), "ACGTNNNN"])
    try:
        with pytest.raises(bridge.SwHipError, match="too long|limit|longer"):
            db.scores_noalloc(_rnd(rng, 70000), s)                      # beyond the stated maximum query length
        with pytest.raises(bridge.SwHipError, match="ACGT"):
            db.scores_noalloc("ACGTNACGT", s)                           # outside the packed kernel's alphabet
        with pytest.raises(bridge.SwHipError, match="ACGT"):
            dbn.scores_noalloc("ACGTACGT", s)
        with pytest.raises(bridge.SwHipError, match="scoring"):
            db.scores_noalloc("ACGTACGT", dlsa_amd.AlignmentScore(2, -1, 0))
        # a query whose boundary column does not fit the fixed ring even once
        with pytest.raises(bridge.SwHipError, match="too long"):
            db.scores_noalloc(_rnd(rng, 60000), s)
        # the library is still usable afterwards
        q = _rnd(rng, 100)
        assert np.array_equal(db.scores(q, s), db.scores_noalloc(q, s))
        assert list(db.scores_noalloc("", s)) == [0] * 200
    finally:
        db.close()
        dbn.close()
