"""Committed golden fixtures (tests/golden/oracle_fixtures.json, made by tests/golden/make_fixtures.py with
the oracle and cross-checked with the reference's own verifier master/utils/verify.py in the build container).
CPU: the oracle still reproduces them.  GPU: the C ABI reproduces them bit for bit."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fixtures():
    with open(os.path.join(HERE, "golden", "oracle_fixtures.json")) as f:
        return json.load(f)["cases"]


def test_oracle_reproduces_golden_fixtures(oracle, fixtures):
    for c in fixtures:
        s = c["scores"]
        for func, e in c["expected"].items():
            r = oracle.align(func, c["query"], c["target"], s["gap"], s["match"], s["miss"])
            assert (r["q"], r["t"], r["score"], r["max_x"], r["max_y"]) == (
                e["q_aln"], e["t_aln"], e["score"], e["max_x"], e["max_y"]), (c["tag"], func)


@pytest.mark.gpu
def test_abi_reproduces_golden_fixtures(fixtures):
    import dlsa_amd
    entry = {"lowmem": dlsa_amd.find_rust_alignment_simd_low_mem, "simd": dlsa_amd.find_rust_alignment_simd,
             "straight": dlsa_amd.find_rust_alignment_sequential}
    for c in fixtures:
        s = c["scores"]
        sc = dlsa_amd.AlignmentScore(s["match"], s["miss"], s["gap"])
        for func, e in c["expected"].items():
            g, err = entry[func](c["query"], c["target"], sc)
            assert err is None, (c["tag"], func, err)
            assert (g.query, g.target, g.score, g.max_x, g.max_y) == (
                e["q_aln"], e["t_aln"], e["score"], e["max_x"], e["max_y"]), (c["tag"], func)
