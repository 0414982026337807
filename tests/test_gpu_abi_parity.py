"""GPU parity of the drop-in C ABI (include/sw.h) against the oracle: every known-answer vector of the
reference, random pairs on all three entry points (bit-exact strings, score, max_x, max_y), the quirk
inputs, ownership rules.  Everything goes through the C ABI via dlsa_amd.bridge (ctypes)."""
import random

import pytest

pytestmark = pytest.mark.gpu

SCORES = [(-1, 2, -1), (-1, 3, -1), (-3, 5, -2), (-2, 1, -1), (-1, 1, -1), (-2, 3, -3), (-5, 4, -1)]
ENTRY = {"lowmem": "find_rust_alignment_simd_low_mem", "simd": "find_rust_alignment_simd",
         "straight": "find_rust_alignment_sequential"}


def _gpu(func, q, t, sc):
    import dlsa_amd
    gap, match, miss = sc
    res, err = getattr(dlsa_amd, ENTRY[func])(q, t, dlsa_amd.AlignmentScore(match, miss, gap))
    assert err is None, err
    return res


def _check(oracle, func, q, t, sc):
    exp = oracle.align(func, q, t, *sc)
    got = _gpu(func, q, t, sc)
    assert exp is not None
    assert (got.query, got.target, got.score, got.max_x, got.max_y) == (
        exp["q"], exp["t"], exp["score"] & 0xFFFF, exp["max_x"], exp["max_y"]), (func, q, t, sc)


def _rnd(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def test_known_answer_vectors(oracle, kat):
    for v in kat["rust_lib_rs"]:
        s = v["scores"]
        sc = (s["gap"], s["match"], s["miss"])
        for func in ENTRY:
            got = _gpu(func, v["query"], v["target"], sc)
            assert (got.query, got.target) == (v["q_aln"], v["t_aln"]), (func, v["id"])
            _check(oracle, func, v["query"], v["target"], sc)
    for v in kat["master_tests_data"]:
        s = v["scores"]
        got = _gpu("lowmem", v["query"], v["target"], (s["gap"], s["match"], s["miss"]))
        assert (got.query, got.target, got.score, got.max_x, got.max_y) == (
            v["q_aln"], v["t_aln"], v["score"], v["max_x"], v["max_y"])


def test_fallback_chain_like_the_go_worker(oracle, kat):
    import dlsa_amd
    for v in kat["rust_lib_rs"][:12]:
        s = v["scores"]
        res, err = dlsa_amd.find_alignment_with_fallback(v["query"], v["target"],
                                                         dlsa_amd.AlignmentScore(s["match"], s["miss"], s["gap"]))
        assert err is None and (res.query, res.target) == (v["q_aln"], v["t_aln"])


@pytest.mark.parametrize("seed", [21, 22])
def test_random_pairs_all_entries(oracle, seed):
    rng = random.Random(seed)
    for _ in range(120):
        mode = rng.random()
        alpha = rng.choice(["AC", "ACGT", "ACGT", "A"])
        if mode < 0.3:
            Q, T = rng.randint(0, 40), rng.randint(0, 100)
        elif mode < 0.5:  # ring path with T < Qp
            Q = rng.randint(1, 31)
            T = rng.randint(2 * Q, 70)
        elif mode < 0.75:  # full path incl. T % 64 == 0
            Q = rng.randint(30, 600)
            T = rng.choice([64, 128, 192, 256, 320, rng.randint(1, 700)])
        else:  # ring path, several strips / wrap
            Q = rng.randint(1, 300)
            T = rng.randint(2 * Q, 2 * Q + 900)
        q, t = _rnd(rng, Q, alpha), _rnd(rng, T, alpha)
        if rng.random() < 0.4 and Q > 3 and T > Q:
            p = rng.randint(0, T - 1)
            t = (t[:p] + q[:rng.randint(1, Q)] + t[p:])[:T]
        sc = rng.choice(SCORES)
        for func in ENTRY:
            _check(oracle, func, q, t, sc)


def test_quirk_inputs(oracle):
    q = "G" * 80 + "TTTTTTTTTACGTA"
    t = "C" * 59 + "ACGTA"
    _check(oracle, "simd", q, t, (-1, 2, -1))
    _check(oracle, "lowmem", q, t, (-1, 2, -1))
    t2 = "C" * 63 + "A"
    _check(oracle, "simd", "A" + "G" * 20, t2, (-1, 2, -1))   # H(1,T) is the only hit
    _check(oracle, "simd", "A" + "G" * 80, t2, (-1, 2, -1))   # ... and is not computed when Tp < Qp
    _check(oracle, "lowmem", "A" + "G" * 20, t2, (-1, 2, -1))
    for func in ENTRY:  # no hit / empty inputs
        for q, t in (("AAAA", "TTTT"), ("", "ACGT"), ("ACGT", ""), ("", "")):
            _check(oracle, func, q, t, (-1, 2, -1))


def test_longer_pairs_many_strips(oracle):
    rng = random.Random(7)
    q = _rnd(rng, 1500)
    core = q[200:1300]
    mutated = "".join(c if rng.random() > 0.08 else rng.choice("ACGT") for c in core)
    t = _rnd(rng, 700) + mutated + _rnd(rng, 500)
    for func in ENTRY:
        _check(oracle, func, q, t, (-1, 2, -1))
    t_long = _rnd(rng, 5000) + q[300:900] + _rnd(rng, 3000)   # ring path, T >> data_store_height
    _check(oracle, "lowmem", q[300:1000], t_long, (-1, 2, -1))
    _check(oracle, "lowmem", q, q, (-1, 2, -1))               # identical sequences


def test_non_ascii_and_arbitrary_alphabet(oracle):
    _check(oracle, "lowmem", "xxabcxx", "zzzzzabczzzzzzzzzzzz", (-1, 2, -1))
    _check(oracle, "simd", "héllo wörld", "say héllo wörld now", (-1, 2, -1))
    _check(oracle, "lowmem", "中文", "abc中文def", (-1, 2, -1))


def test_invalid_utf8_and_gap_zero_return_null():
    import dlsa_amd
    from dlsa_amd import bridge
    L = bridge.lib()
    sc = dlsa_amd.AlignmentScore(2, -1, -1).c()
    assert not L.find_alignment_low_memory(b"\xff\xfe", b"ACGT", sc)
    assert not L.find_alignment_simd(b"ACGT", b"\xc3", sc)
    # rust/src/algorithm/mod.rs:236-238: gap == 0 divides by zero on the ring path -> NULL, simd still works
    z = dlsa_amd.AlignmentScore(2, -1, 0)
    res, err = dlsa_amd.find_rust_alignment_simd_low_mem("AC", "ACGTACGT", z)
    assert res is None and err is not None
    res, err = dlsa_amd.find_alignment_with_fallback("AC", "ACGTACGT", z)
    assert err is None
    assert not L.find_alignment_parallel(b"A", b"A", 4, sc)
    assert not L.find_alignment_sequential(b"A", b"A", sc)


def test_ownership_rules():
    import ctypes as C
    import dlsa_amd
    from dlsa_amd import bridge
    L = bridge.lib()
    sc = dlsa_amd.AlignmentScore(2, -1, -1).c()
    L.free_alignment_result(None)
    L.free_c_string(None)
    r = L.find_alignment_low_memory(b"ACGT", b"ACGT", sc)
    assert C.string_at(r.contents.query_ptr) == b"ACGT"
    L.free_c_string(r.contents.query_ptr)      # either string may be freed first ...
    L.free_alignment_result(r)                 # ... and the struct afterwards, exactly once
    r = L.find_alignment_simd(b"ACGT", b"ACGT", sc)
    L.free_c_string(r.contents.target_ptr)
    L.free_c_string(r.contents.query_ptr)
    L.free_alignment_result(r)


def test_concurrent_callers(oracle):
    # the Go worker enters from NumCPU-1 OS threads (worker_impl.go:204-238)
    import threading
    rng = random.Random(3)
    pairs = [(_rnd(rng, rng.randint(5, 120)), _rnd(rng, rng.randint(5, 400))) for _ in range(64)]
    exp = [oracle.align("lowmem", q, t, -1, 2, -1) for q, t in pairs]
    got = [None] * len(pairs)

    def work(k):
        for i in range(k, len(pairs), 8):
            got[i] = _gpu("lowmem", pairs[i][0], pairs[i][1], (-1, 2, -1))
    th = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    for e, g in zip(exp, got):
        assert (g.query, g.target, g.score, g.max_x, g.max_y) == (e["q"], e["t"], e["score"], e["max_x"], e["max_y"])


def test_concurrent_packages_searches_and_single_calls(oracle):
    """Everything at once from several threads: two work packages big enough for the two-in-flight driver, database
    searches, and one-pair calls.  The library serialises submissions per device and shares pools (workspaces, ops
    arenas, slabs); every caller must still get exactly what it gets alone."""
    import ctypes as C
    import threading
    import dlsa_amd
    from dlsa_amd import bridge
    L = bridge.lib()
    rng = random.Random(77)
    s = dlsa_amd.AlignmentScore(2, -1, -1)
    n = 36000
    packs = []
    for p in range(2):
        qs = [_rnd(rng, 30 + 10 * p).encode() for _ in range(5)]
        ts = [_rnd(rng, rng.randint(20, 70)).encode() for _ in range(n)]
        packs.append((qs, ts))
    targets = [_rnd(rng, rng.randint(50, 400)) for _ in range(600)]
    db = bridge.Database(seqs=targets)
    dq = _rnd(rng, 120)
    singles = [(_rnd(rng, rng.randint(5, 90)), _rnd(rng, rng.randint(5, 200))) for _ in range(40)]

    def run_pack(p):
        qs, ts = packs[p]
        qarr = (C.c_char_p * n)(*[qs[i % 5] for i in range(n)])
        tarr = (C.c_char_p * n)(*ts)
        out = (C.POINTER(bridge.CResult) * n)()
        assert L.sw_align_batch(qarr, tarr, n, s.c(), 0, out) == 0
        got = [(C.string_at(out[i].contents.query_ptr), C.string_at(out[i].contents.target_ptr), out[i].contents.score,
                out[i].contents.max_x, out[i].contents.max_y) for i in range(0, n, 53)]
        L.sw_free_alignment_results(out, n)
        return got

    def run_search():
        return [[(h["target_index"], h["score"], h["query_aln"], h["target_aln"]) for h in db.search_topk(dq, s, 7)] for _ in range(6)]

    def run_singles():
        return [_gpu("lowmem", q, t, (-1, 2, -1)) for q, t in singles]

    try:
        alone = [run_pack(0), run_pack(1), run_search(), run_singles()]
        res = [None] * 4
        jobs = [lambda: run_pack(0), lambda: run_pack(1), run_search, run_singles]
        errs = []

        def work(k):
            try:
                res[k] = jobs[k]()
            except BaseException as e:  # noqa: BLE001 -- reported below, in the main thread
                errs.append((k, repr(e)))
        th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        assert res[0] == alone[0] and res[1] == alone[1] and res[2] == alone[2]
        for a, b in zip(res[3], alone[3]):
            assert (a.query, a.target, a.score, a.max_x, a.max_y) == (b.query, b.target, b.score, b.max_x, b.max_y)
        # ... and what they get alone is the oracle's (sampled)
        qs, ts = packs[0]
        for k, i in enumerate(range(0, n, 53)):
            if k % 40:
                continue
            e = oracle.align("lowmem", qs[i % 5].decode(), ts[i].decode(), -1, 2, -1)
            assert (alone[0][k][0].decode(), alone[0][k][1].decode(), alone[0][k][2]) == (e["q"], e["t"], e["score"] & 0xFFFF)
    finally:
        db.close()


def test_batch_results_freed_in_any_mix(oracle):
    """The results of one sw_align_batch call share a slab: every way of freeing them -- one by one, strings first, in
    bulk (sw_free_alignment_results), or a mix -- releases each exactly once; a later batch of the same shape (which
    may be handed the recycled slab) still comes out right."""
    import ctypes as C
    import dlsa_amd
    from dlsa_amd import bridge
    L = bridge.lib()
    rng = random.Random(12)
    n = 40000                                            # > 32768 pairs: sub-batches, two in flight, a slab each
    qs = [_rnd(rng, 40).encode() for _ in range(8)]
    ts = [_rnd(rng, rng.randint(20, 60)).encode() for _ in range(n)]
    qarr = (C.c_char_p * n)(*[qs[i % 8] for i in range(n)])
    tarr = (C.c_char_p * n)(*ts)
    sc = dlsa_amd.AlignmentScore(2, -1, -1).c()
    first = None
    for round_ in range(3):
        out = (C.POINTER(bridge.CResult) * n)()
        assert L.sw_align_batch(qarr, tarr, n, sc, 0, out) == 0
        got = [(C.string_at(out[i].contents.query_ptr), out[i].contents.score) for i in range(0, n, 997)]
        if first is None:
            first = got
            for i in range(0, n, 211):                   # ... and every sub-batch gives the oracle's results
                r = out[i].contents
                e = oracle.align("lowmem", qs[i % 8].decode(), ts[i].decode(), -1, 2, -1)
                assert (C.string_at(r.query_ptr).decode(), C.string_at(r.target_ptr).decode(), r.score, r.max_x, r.max_y) == (
                    e["q"], e["t"], e["score"] & 0xFFFF, e["max_x"] % 2**64, e["max_y"] % 2**64), i
        assert got == first
        for i in range(0, n, 3):                         # a third the reference's way: one call per result,
            if i % 2:
                L.free_c_string(out[i].contents.target_ptr)   # some with a string freed first
            L.free_alignment_result(out[i])
            out[i] = None
        L.sw_free_alignment_results(out, n)              # the rest in bulk (NULL entries are skipped)
        assert all(not out[i] for i in range(n))
        L.sw_free_alignment_results(out, n)              # and again: nothing left to free


def test_align_batch_matches_single_calls(oracle):
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(9)
    qs = [_rnd(rng, rng.randint(0, 300)) for _ in range(200)]
    ts = [_rnd(rng, rng.randint(0, 600)) for _ in range(200)]
    for path, func in ((0, "lowmem"), (1, "simd"), (2, "straight")):
        res = bridge.align_batch(qs, ts, dlsa_amd.AlignmentScore(2, -1, -1), path)
        for q, t, g in zip(qs, ts, res):
            e = oracle.align(func, q, t, -1, 2, -1)
            assert (g.query, g.target, g.score, g.max_x, g.max_y) == (e["q"], e["t"], e["score"], e["max_x"], e["max_y"])


def _verify_alignment(q, t, res, sc):
    """master/utils/verify.py:24-48 restated: the aligned strings re-derive the score and end at max_x/max_y."""
    gap, match, miss = sc
    s = 0
    for a, b in zip(res.query, res.target):
        s += gap if (a == "-" or b == "-") else (match if a == b else miss)
    assert s == res.score32
    qa, ta = res.query.replace("-", ""), res.target.replace("-", "")
    assert q[res.max_x + 1 - len(qa):res.max_x + 1] == qa
    assert t[res.max_y + 1 - len(ta):res.max_y + 1] == ta


def test_long_pairs_strips_chained_across_the_gpu(oracle):
    """Several thousand rows = tens of 256-row strips, each on its own wave somewhere on the GPU; compared with
    the clean int32 model (the line-by-line restatement's full matrix would be GBs here).  Scores above 32767
    are the documented int32 divergence: Result.score holds the low 16 bits, sw_result_score32 the value."""
    import dlsa_amd
    rng = random.Random(99)
    sc = (-1, 2, -1)
    cases = []
    q = _rnd(rng, 6000)
    cases.append((q, _rnd(rng, 7000)))                                   # unrelated: low score, long matrix
    mut = "".join(c if rng.random() > 0.05 else rng.choice("ACGT") for c in q[500:5500])
    cases.append((q, _rnd(rng, 300) + mut + _rnd(rng, 900)))             # 5 kb hit, score < 32767
    q2 = _rnd(rng, 20000)
    mut2 = "".join(c if rng.random() > 0.03 else rng.choice("ACGT") for c in q2)
    cases.append((q2, mut2[:19990]))                                      # score > 32767: int32
    cases.append((_rnd(rng, 9000), _rnd(rng, 64 * 40)))                   # T % 64 == 0 quirk, full path
    cases.append((_rnd(rng, 700), _rnd(rng, 30000)))                      # ring path, long target
    for q, t in cases:
        exp = oracle.clean_align(0, q, t, *sc)
        res, err = dlsa_amd.find_rust_alignment_simd_low_mem(q, t, dlsa_amd.AlignmentScore(sc[1], sc[2], sc[0]))
        assert err is None, err
        assert (res.score32, res.max_x, res.max_y) == (exp["score"], exp["max_x"], exp["max_y"]), (len(q), len(t))
        assert (res.query, res.target) == (exp["q"], exp["t"])
        assert res.score == exp["score"] & 0xFFFF
        _verify_alignment(q, t, res, sc)


@pytest.mark.gpu
def test_pair_whose_score_could_leave_the_kernels_range_is_refused():
    """The exact kernel keeps 4 * H in an int32 whose top bit is the hand-over tag (sw_exact.hip "arithmetic of one
    cell"): a pair that could score 2^29 is an error, not a wrong answer.  (The reference computes in i16 and is
    undefined far below that: rust/src/algorithm/mod.rs:66-202.)"""
    import dlsa_amd
    from dlsa_amd import bridge
    sc = dlsa_amd.AlignmentScore(32767, -1, -1)
    with pytest.raises(bridge.SwHipError, match="2\\^29"):
        bridge.align_batch([b"A" * 9000], [b"A" * 9000], sc, 0)
    # just inside: 2 * 8001 * 32767 + 65536 < 2^29, and the score is exact in 32 bits (one match per aligned column)
    r = bridge.align_batch([b"A" * 8001], [b"A" * 8001], sc, 0)[0]
    assert r is not None and len(r.query) >= 7900 and r.query == r.target
    assert r.score32 == len(r.query) * 32767 and r.score == r.score32 & 0xffff
