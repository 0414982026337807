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
