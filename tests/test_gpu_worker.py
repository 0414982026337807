"""GPU parity of the worker-side callers (include/sw_worker.h via dlsa_amd/worker.py) against the oracle driven
like the Go worker: ExecuteWork over a work package (fallback chain, sign flip, int() conversions), the capacity
benchmark's protocol, and the CLI's FASTA -> top-k result files."""
import json
import random

import pytest

pytestmark = pytest.mark.gpu


def _rnd(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def _go_int(v):  # int(uint64) in Go on a 64-bit platform
    return v - (1 << 64) if v >= (1 << 63) else v


def _package(rng, n_q, n_t, scores=(2, 1, 1)):
    from dlsa_amd import worker
    seqs = {}
    for i in range(n_q):
        seqs[f"q{i}"] = _rnd(rng, rng.randint(20, 300))
    for i in range(n_t):
        seqs[f"t{i}"] = _rnd(rng, rng.randint(1, 700))
    seqs["t0"] = seqs["t0"][:10] + seqs["q0"][5:150] + seqs["t0"][10:]
    seqs["t1"] = ""                      # empty target: no hit, coordinates -1 after Go's int()
    seqs["t2"] = "héllo wörld " + seqs["q1"][:15]  # non-ASCII: byte length vs char count
    seqs["q1"] = seqs["q1"][:15] + " wörld"
    pairs = [worker.QueryTargetType(f"q{a}", f"t{b}") for a in range(n_q) for b in range(n_t)]
    rng.shuffle(pairs)
    return worker.CompleteWorkPackage("wp", "job", pairs, scores[0], scores[1], scores[2], seqs)


def _expected(oracle, work, comb):
    e = oracle.align_with_fallback(work.sequences[comb.query], work.sequences[comb.target],
                                   -work.gap_penalty, work.match_score, -work.mismatch_penalty)
    assert e is not None
    return e


def test_execute_work_matches_the_go_worker_on_the_oracle(oracle):
    from dlsa_amd import worker
    rng = random.Random(31)
    work = _package(rng, 6, 45)
    w = worker.Worker()
    w.execute_work(work, work.queries)
    details = [d for r in w.sent for d in r.alignments]
    assert len(details) == len(work.queries)
    assert [len(r.alignments) for r in w.sent][:2] == [101, 100]
    for comb, d in zip(work.queries, details):
        e = _expected(oracle, work, comb)
        a = d.alignment
        assert d.combination == {"target": comb.target, "query": comb.query}
        assert (a.query_alignment, a.target_alignment) == (e["q"], e["t"]), (comb,)
        assert a.score == e["score"] & 0xFFFF
        assert (a.maxX, a.maxY) == (_go_int(e["max_x"]), _go_int(e["max_y"]))
        assert a.length == len(e["q"].encode("utf-8"))      # Go len() counts bytes
        assert d.entry == e["which"]
    # a body the master accepts
    body = json.loads(w.sent[0].to_json())
    assert set(body["alignments"][0]["alignment"]) == {"query_alignment", "target_alignment", "length", "score", "maxX", "maxY"}


def test_execute_work_fallback_chain_gap_zero(oracle):
    """gap penalty 0: find_alignment_low_memory fails on the ring path (division by zero, algorithm/mod.rs:236-238)
    and the worker falls back to find_alignment_simd (worker_impl.go:184)."""
    from dlsa_amd import worker
    seqs = {"q": "ACGT", "t_ring": "TTTTACGTTTTTTTTT", "t_full": "ACGGT"}
    pairs = [worker.QueryTargetType("q", "t_ring"), worker.QueryTargetType("q", "t_full")]
    work = worker.CompleteWorkPackage("wp", "job", pairs, 2, 1, 0, seqs)
    det = worker.Worker().align(work, pairs)
    assert [d.entry for d in det] == [1, 0]
    for comb, d in zip(pairs, det):
        e = _expected(oracle, work, comb)
        assert (d.alignment.query_alignment, d.alignment.target_alignment, d.alignment.score, d.entry) == (
            e["q"], e["t"], e["score"] & 0xFFFF, e["which"])
    with pytest.raises(worker.SwHipError):
        worker.Worker().align(work, [worker.QueryTargetType("q", "missing")])


def test_capacity_benchmarks():
    """BenchmarkRust's protocol on one pair, and the batch capacity figure: positive, and the registered number is
    integer-valued (the master's field is an int, master/api_models/worker.py:7-16)."""
    from dlsa_amd import worker
    one_pair = worker.benchmark_rust(0.010, 1, 2)
    assert one_pair > 1e9            # the reference's CPU figure is ~1e9..1e10 on one core
    cap = worker.benchmark_capacity()
    assert cap > 1e11 and cap == int(cap)
    body = json.loads(worker.machine_specs_request(cap))
    assert float(body["benchmark_result"]).is_integer()


def test_fasta_to_result_files(oracle, tmp_path):
    """`python -m cli --query Q --database D --top-k 3` without the coordinator: per query file, the CLI's format,
    hits ordered by score (ties: database order)."""
    from dlsa_amd import worker
    rng = random.Random(41)
    queries = {"qA": _rnd(rng, 120), "qB": _rnd(rng, 80)}
    targets = {f"seq{i}": _rnd(rng, rng.randint(60, 300)) for i in range(40)}
    targets["seq7"] = targets["seq7"][:30] + queries["qA"][10:100] + targets["seq7"][30:]
    targets["seq9"] = queries["qB"][5:70]
    qf, tf = tmp_path / "q.fasta", tmp_path / "db.fasta"
    qf.write_text("".join(f">{k} description\n{v[:50]}\n{v[50:]}\n" for k, v in queries.items()))
    tf.write_text("".join(f">{k}\n{v}\n" for k, v in targets.items()))
    out = tmp_path / "results"
    worker.search_fasta(str(qf), str(tf), 2, 1, 1, top_k=3, output_path=str(out))
    names = list(targets)
    for qname, q in queries.items():
        exp = [oracle.align("lowmem", q, targets[t], -1, 2, -1) for t in names]
        order = sorted(range(len(names)), key=lambda i: (-exp[i]["score"], i))[:3]
        want = "".join(f">{names[i]}\nAlignment: {exp[i]['q']}\nLength: {len(exp[i]['q'])}\nScore: {exp[i]['score']}\n\n"
                       for i in order)
        assert (out / f"{qname}.txt").read_text() == want


def test_command_line_drivers(oracle, tmp_path, capsys):
    """`python -m dlsa_amd work / search / benchmark`: a CompleteWorkPackage file in, the bodies the worker would POST
    out (one JSON per line, 101 + 100 + ... results like worker_impl.go:158-166); the CLI's flags without the server."""
    from dlsa_amd import __main__ as cli
    rng = random.Random(52)
    work = _package(rng, 3, 60)
    pkg = {"id": work.id, "job_id": work.job_id, "match_score": work.match_score, "mismatch_penalty": work.mismatch_penalty,
           "gap_penalty": work.gap_penalty, "sequences": work.sequences,
           "queries": [{"query": c.query, "target": c.target} for c in work.queries]}
    pf, of = tmp_path / "package.json", tmp_path / "posts.jsonl"
    pf.write_text(json.dumps(pkg))
    assert cli.main(["work", str(pf), "--out", str(of), "--cpus", "1"]) == 0
    posts = [json.loads(line) for line in of.read_text().splitlines()]
    assert [len(p["alignments"]) for p in posts] == [101, 79]
    details = [d for p in posts for d in p["alignments"]]
    for comb, d in zip(work.queries, details):
        e = _expected(oracle, work, comb)
        assert d["combination"] == {"target": comb.target, "query": comb.query}
        a = d["alignment"]
        assert (a["query_alignment"], a["target_alignment"], a["score"]) == (e["q"], e["t"], e["score"] & 0xFFFF)
        assert (a["maxX"], a["maxY"]) == (_go_int(e["max_x"]), _go_int(e["max_y"]))
    # the same package over a node handle (all GPUs of the box)
    of2 = tmp_path / "posts_node.jsonl"
    assert cli.main(["work", str(pf), "--out", str(of2), "--cpus", "1", "--node"]) == 0
    assert of2.read_text() == of.read_text()
    # search: the CLI's flags
    qf, tf = tmp_path / "q.fasta", tmp_path / "db.fasta"
    q = _rnd(rng, 90)
    ts = {f"s{i}": _rnd(rng, rng.randint(40, 200)) for i in range(12)}
    ts["s4"] = ts["s4"][:20] + q[10:80] + ts["s4"][20:]
    qf.write_text(f">query1\n{q}\n")
    tf.write_text("".join(f">{k}\n{v}\n" for k, v in ts.items()))
    out = tmp_path / "res"
    assert cli.main(["search", "--query", str(qf), "--database", str(tf), "--top-k", "2", "--output-path", str(out)]) == 0
    text = (out / "query1.txt").read_text()
    e = oracle.align("lowmem", q, ts["s4"], -1, 2, -1)
    assert text.startswith(f">s4\nAlignment: {e['q']}\nLength: {len(e['q'])}\nScore: {e['score']}\n\n") and text.count(">") == 2
    # benchmark: the registration body, integer-valued
    capsys.readouterr()
    assert cli.main(["benchmark"]) == 0
    body = json.loads(capsys.readouterr().out)
    assert body["benchmark_result"] > 1e11 and body["benchmark_result"] == float(int(body["benchmark_result"]))
