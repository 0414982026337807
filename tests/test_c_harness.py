"""The plain-C worker harness (tests/c_harness/worker_harness.c): the public headers compile as C, the library links
with the cgo link line of INTEGRATION.md, and -- on the GPU -- N threads calling the drop-in ABI one pair at a time
(the Go worker's real call pattern, worker_impl.go:204-238) get the oracle's answers."""
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS_DIR = os.path.join(ROOT, "tests", "c_harness")
HARNESS = os.path.join(HARNESS_DIR, "worker_harness")


def _build():
    from dlsa_amd import build
    build.build()
    subprocess.check_call(["make", "-C", HARNESS_DIR], stdout=subprocess.DEVNULL)


def test_harness_compiles_as_c_links_and_selftests():
    _build()
    out = subprocess.run([HARNESS, "selftest"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "selftest ok" in out.stdout


@pytest.mark.gpu
def test_threads_calling_one_pair_at_a_time_match_the_oracle(oracle, tmp_path):
    if not os.path.exists(HARNESS):
        _build()
    rng = random.Random(77)
    rnd = lambda n, a="ACGT": "".join(rng.choice(a) for _ in range(n))  # noqa: E731
    scorings = [(-1, 2, -1), (-2, 1, -1), (-3, 5, -2), (0, 2, -1)]
    pairs = []
    for i in range(180):
        gap, match, miss = scorings[i % len(scorings)]
        Q = rng.choice([0, 1, 5, 31, 64, 100, 257, 600])
        T = rng.choice([0, 3, 63, 64, 128, 200, 700, 1500])
        q, t = rnd(Q), rnd(T)
        if Q > 20 and T > Q:
            t = (t[:T // 3] + q[2:Q - 2] + t[T // 3:])[:T]
        pairs.append((gap, match, miss, q, t))
    f = tmp_path / "pairs.txt"
    f.write_text("".join(f"{g} {m} {x} {q or '-'} {t or '-'}\n" for g, m, x, q, t in pairs))
    out = subprocess.run([HARNESS, "pairs", str(f), "7"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.split("\n")[:-1]
    assert len(lines) == len(pairs)
    for (gap, match, miss, q, t), line in zip(pairs, lines):
        entry, score, mx, my, qlen, qa, ta = line.split("\t")
        e = oracle.align_with_fallback(q, t, gap, match, miss)
        go_int = lambda v: v - (1 << 64) if v >= (1 << 63) else v  # noqa: E731
        assert (int(entry), int(score), int(mx), int(my), int(qlen), qa, ta) == (
            e["which"], e["score"] & 0xFFFF, go_int(e["max_x"]), go_int(e["max_y"]), len(e["q"]), e["q"], e["t"]), (len(q), len(t))
