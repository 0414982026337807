"""The oracle (oracle/sw_oracle.c) against every known-answer vector the reference's own tests hold for
this path.  This is what pins the restatement; nothing here touches the GPU."""
import pytest

FUNCS = ["sequential", "straight", "simd", "lowmem"]  # rust/src/lib.rs:423-457


@pytest.mark.parametrize("func", FUNCS)
def test_rust_known_answers(oracle, kat, func):
    for v in kat["rust_lib_rs"]:
        s = v["scores"]
        r = oracle.align(func, v["query"], v["target"], s["gap"], s["match"], s["miss"])
        assert r is not None, (func, v["id"])
        assert (r["q"], r["t"]) == (v["q_aln"], v["t_aln"]), (func, v["id"])


def test_fallback_chain_known_answers(oracle, kat):
    # the Go worker's entry (internal/worker/worker_impl.go:170-202) must give the same strings and
    # must be served by the first entry (no panic on any known-answer vector)
    for v in kat["rust_lib_rs"]:
        s = v["scores"]
        r = oracle.align_with_fallback(v["query"], v["target"], s["gap"], s["match"], s["miss"])
        assert r["which"] == 0
        assert (r["q"], r["t"]) == (v["q_aln"], v["t_aln"]), v["id"]


def test_go_vector(oracle, kat):
    for v in kat["go_smithwaterman_test"]:
        s = v["scores"]
        for func in ("simd", "lowmem", "straight"):
            r = oracle.align(func, v["query"], v["target"], s["gap"], s["match"], s["miss"])
            assert r["q"] == v["q_aln"], func


def test_protocol_fixture_coordinates(oracle, kat):
    # master/tests/data.py:43-63: score 4, maxX = maxY = 3 for ABCD/ABCD with match 1
    for v in kat["master_tests_data"]:
        s = v["scores"]
        for func in ("simd", "lowmem"):
            r = oracle.align(func, v["query"], v["target"], s["gap"], s["match"], s["miss"])
            assert (r["q"], r["t"], r["score"], r["max_x"], r["max_y"]) == (
                v["q_aln"], v["t_aln"], v["score"], v["max_x"], v["max_y"])


def test_no_hit_convention(oracle):
    # x = 0 -> x - 1 wraps (rust/src/lib.rs:135, algorithm/mod.rs:444); straight returns (0, 0) (lib.rs:70)
    for func in ("simd", "lowmem", "sequential"):
        r = oracle.align(func, "AAAA", "TTTT", -1, 2, -1)
        assert (r["q"], r["t"], r["score"], r["max_x"], r["max_y"]) == ("", "", 0, oracle.U64MAX, oracle.U64MAX)
    r = oracle.align("straight", "AAAA", "TTTT", -1, 2, -1)
    assert (r["q"], r["t"], r["score"], r["max_x"], r["max_y"]) == ("", "", 0, 0, 0)
    for q, t in (("", "ACGT"), ("ACGT", ""), ("", "")):
        for func in ("simd", "lowmem"):
            r = oracle.align(func, q, t, -1, 2, -1)
            assert (r["q"], r["score"], r["max_x"]) == ("", 0, oracle.U64MAX)


def test_gap_zero_panics_on_ring_only(oracle):
    # algorithm/mod.rs:236-238 divides by |gap| on the ring path only
    assert oracle.align("lowmem", "AC", "ACGTACGT", 0, 2, -1) is None
    assert oracle.align("simd", "AC", "ACGTACGT", 0, 2, -1) is not None
    r = oracle.align_with_fallback("AC", "ACGTACGT", 0, 2, -1)
    assert r["which"] == 1
