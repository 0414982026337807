"""Failure containment of the exact kernel's strip pipeline (dlsa_amd/csrc/sw_exact.hip): a bottom row that is never
published must end in an error within the wait budget -- never in a result computed from an incomplete boundary, never
in a hang -- and the library must work again afterwards.  SW_HIP_DEBUG_DROP_HANDOVER makes strip 0 of the first pair
skip its hand-over stores; SW_HIP_EXACT_WAIT_MS shortens the budget (default 4 s).  Both are read per submission."""
import os
import random
import time

import pytest

pytestmark = pytest.mark.gpu


def test_unpublished_boundary_is_reported_within_the_deadline(oracle):
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(5)
    q = "".join(rng.choice("ACGT") for _ in range(900))
    t = "".join(rng.choice("ACGT") for _ in range(900))
    sc = dlsa_amd.AlignmentScore(2, -1, -1)
    good, err = dlsa_amd.find_rust_alignment_simd(q, t, sc)
    assert err is None
    os.environ["SW_HIP_DEBUG_DROP_HANDOVER"] = "1"
    os.environ["SW_HIP_EXACT_WAIT_MS"] = "300"
    try:
        t0 = time.perf_counter()
        res, err = dlsa_amd.find_rust_alignment_simd(q, t, sc)
        dt = time.perf_counter() - t0
        assert res is None and err is not None, "a result was computed from an incomplete boundary"
        assert "timed out" in bridge.last_error()
        assert dt < 3.0, dt
        # a batch whose first pair is hit fails as a whole (nothing of the submission may be used)
        with pytest.raises(bridge.SwHipError):
            bridge.align_batch([q, q[:300]], [t, t[:300]], sc, 1)
    finally:
        os.environ.pop("SW_HIP_DEBUG_DROP_HANDOVER", None)
        os.environ.pop("SW_HIP_EXACT_WAIT_MS", None)
    again, err = dlsa_amd.find_rust_alignment_simd(q, t, sc)
    assert err is None
    assert (again.query, again.target, again.score, again.max_x, again.max_y) == (
        good.query, good.target, good.score, good.max_x, good.max_y)
    e = oracle.align("simd", q, t, -1, 2, -1)
    assert (again.query, again.target, again.score) == (e["q"], e["t"], e["score"] & 0xFFFF)


def test_failure_on_a_context_lane_fails_the_batch_and_nothing_else(oracle):
    """Many small target sets run side by side on context lanes (sw_package.cpp); a hand-over that never comes on
    one of them must fail the whole batch with its message, leave no lane locked or busy, and the next batch must
    give the same records as before."""
    import dlsa_amd
    from dlsa_amd import bridge
    rng = random.Random(6)
    r = lambda n: "".join(rng.choice("ACGT") for _ in range(n))  # noqa: E731
    qs = [r(rng.randint(500, 900)) for _ in range(6)]
    ts = [r(rng.randint(8000, 20000)) for _ in range(6)]
    sc = dlsa_amd.AlignmentScore(2, -1, -1)
    good = bridge.align_batch(qs, ts, sc, 0)
    assert bridge.package_stats()["pairs"] == 6          # the fast path took them: six sets of one target
    os.environ["SW_HIP_DEBUG_DROP_HANDOVER"] = "1"
    os.environ["SW_HIP_EXACT_WAIT_MS"] = "300"
    try:
        t0 = time.perf_counter()
        with pytest.raises(bridge.SwHipError) as ei:
            bridge.align_batch(qs, ts, sc, 0)
        assert "timed out" in str(ei.value) or "timed out" in bridge.last_error()
        assert time.perf_counter() - t0 < 5.0
    finally:
        os.environ.pop("SW_HIP_DEBUG_DROP_HANDOVER", None)
        os.environ.pop("SW_HIP_EXACT_WAIT_MS", None)
    again = bridge.align_batch(qs, ts, sc, 0)
    key = lambda x: (x.query, x.target, x.score, x.max_x, x.max_y)  # noqa: E731
    assert [key(x) for x in again] == [key(x) for x in good]
    e = oracle.align("lowmem", qs[2], ts[2], -1, 2, -1)
    assert (again[2].query, again[2].target, again[2].score) == (e["q"], e["t"], e["score"] & 0xFFFF)
