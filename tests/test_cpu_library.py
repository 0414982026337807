"""CPU-side checks of the product library (no GPU, no compute calls): it builds for gfx950, loads, exports
every symbol the headers declare, has the reference's struct layouts, refuses to compute without a GPU, and
its path dispatch (host arithmetic) agrees with the oracle's clean model."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from dlsa_amd import build, bridge
    build.build()
    return bridge.lib()


def _declared_symbols():
    names = set()
    for h in ("sw.h", "sw_hip.h", "sw_worker.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b([a-z_][a-z0-9_]*)\s*\(", src):
            if m.group(1) not in ("defined", "sizeof"):
                names.add(m.group(1))
    return names


def test_every_declared_symbol_is_exported(L):
    names = _declared_symbols()
    # the five symbols the Go worker binds (rust/src/bindings.rs:72,116,188,233,243) + the two stubs
    for must in ("find_alignment_low_memory", "find_alignment_simd", "find_alignment_sequential_straight",
                 "free_alignment_result", "free_c_string", "find_alignment_parallel", "find_alignment_sequential"):
        assert must in names
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_struct_layouts_match_the_reference_header():
    from dlsa_amd import bridge
    # rust/header/sw.h:4-18 / rust/src/bindings.rs:9-16: Result is 40 bytes, AlignmentScores 6 bytes
    assert C.sizeof(bridge.CResult) == 40
    assert [getattr(bridge.CResult, f).offset for f in ("query_ptr", "target_ptr", "score", "max_x", "max_y")] == [0, 8, 16, 24, 32]
    assert C.sizeof(bridge.CScores) == 6
    assert [getattr(bridge.CScores, f).offset for f in ("gap", "match", "miss")] == [0, 2, 4]
    s = bridge.AlignmentScore(2, -1, -1).c()   # the worker passes C.ushort(-penalty)
    assert (s.gap, s.match, s.miss) == (0xFFFF, 2, 0xFFFF)


def test_no_cpu_fallback(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from dlsa_amd import bridge
    import dlsa_amd
    res, err = dlsa_amd.find_alignment_with_fallback("ACGT", "ACGT", dlsa_amd.AlignmentScore(2, -1, -1))
    assert res is None and err is not None
    assert "no CPU fallback" in bridge.last_error() or "HIP" in bridge.last_error()
    with pytest.raises(bridge.SwHipError):
        bridge.Database(seqs=["ACGT"])
    L.free_alignment_result(None)
    L.free_c_string(None)


def test_path_dispatch_matches_clean_model(L, oracle):
    # The library's resolve_path (host arithmetic) vs the oracle's clean model on lengths around every
    # boundary: T < 2Q dispatch, T % 64 == 0, Tp >= Qp, T < Qp.
    out = (C.c_uint32 * 4)()

    def params(path, Q, T):
        assert L.sw_hip_path_params(path, Q, T, out) == 0
        return tuple(out)

    def expect(path, Q, T):
        Qp, Tp = (Q + 63) // 64 * 64, (T + 63) // 64 * 64
        if path == 2:
            return (T, T, 0, 1)
        full = path == 1 or T < 2 * Q
        if full:
            if T > 0 and T % 64 == 0:
                return (T - 1, T if Tp >= Qp else T - 1, 0, 0)
            return (T, T, 0, 0)
        if T < Qp:
            return (T, T - 1, Qp, 0)
        return (T, T, 0, 0)

    for Q in (0, 1, 2, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 1000):
        for T in (0, 1, 2, 62, 63, 64, 65, 126, 127, 128, 129, 191, 192, 199, 200, 201, 256, 1999, 2000, 2001, 2048):
            for path in (0, 1, 2):
                assert params(path, Q, T) == expect(path, Q, T), (path, Q, T)
    # ... and the clean model built on the same four integers equals the restatement (spot check here, the
    # exhaustive comparison is tests/test_oracle_clean.py)
    q, t = "A" + "G" * 20, "C" * 63 + "A"
    a, c = oracle.align("simd", q, t, -1, 2, -1), oracle.clean_align(1, q, t, -1, 2, -1)
    assert (a["q"], a["score"], a["max_x"], a["max_y"]) == (c["q"], c["score"], c["max_x"], c["max_y"]) == ("A", 2, 0, 63)
    assert params(1, len(q), len(t)) == (63, 64, 0, 0)   # full path: column 64 only for row 1
    assert params(0, len(q), len(t)) == (64, 64, 0, 0)   # T >= 2Q: ring path, no quirk


def test_sharder_covers_and_balances(L):
    """sw_shard_plan (host arithmetic): every target in exactly one shard, residue totals balanced within 2 % on
    the C3 length distribution (10^6 targets of 200..5000 nt over 8 GPUs), whole lane tiles of
    similar lengths per shard, deterministic, and sane on tiny and degenerate databases."""
    import numpy as np
    from dlsa_amd import bridge
    rng = np.random.default_rng(3)
    lens = rng.integers(200, 5001, size=1_000_000)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    for shards in (2, 3, 8):
        so = bridge.shard_plan(offs, shards)
        assert so.min() == 0 and so.max() == shards - 1
        tot = np.bincount(so, weights=lens, minlength=shards)
        assert (tot.max() - tot.min()) / tot.mean() < 0.02, tot
        cnt = np.bincount(so, minlength=shards)
        assert cnt.sum() == len(lens)
        assert np.array_equal(so, bridge.shard_plan(offs, shards))          # deterministic
        # every shard holds targets from every length bucket (a shard of only-short targets would finish early)
        for s in range(shards):
            mine = lens[so == s]
            assert mine.min() < 300 and mine.max() > 4900
        # tiles: in length-sorted order, runs of 128 consecutive targets share a shard
        order = np.argsort(-lens, kind="stable")
        runs = so[order][: (len(lens) // 128) * 128].reshape(-1, 128)
        assert (runs == runs[:, :1]).all()
    # tiny and degenerate inputs
    assert list(bridge.shard_plan(np.array([0], dtype=np.uint64), 4)) == []
    one = bridge.shard_plan(np.array([0, 10], dtype=np.uint64), 3)
    assert list(one) == [0]
    offs = np.arange(0, 70, 10, dtype=np.uint64)                               # 6 equal targets over 3 shards
    so = bridge.shard_plan(offs, 3)
    assert sorted(np.bincount(so, minlength=3)) == [2, 2, 2]
    zeros = np.zeros(11, dtype=np.uint64)                                       # empty sequences still spread
    assert len(set(bridge.shard_plan(zeros, 2))) == 2
    bad = np.array([0, 5, 3], dtype=np.uint64)
    with pytest.raises(bridge.SwHipError):
        bridge.shard_plan(bad, 2)


def test_score_unit_tables_keep_their_invariants(L):
    """The ticket-ordered unit table of the score kernel (host arithmetic, sw_hip_test_units): every strip exactly
    once, dependencies and ring-slot reuse point to EARLIER tickets (that is what makes the cross-workgroup chain
    deadlock-free), and the partial last round is cut into short units."""
    import random
    rng = random.Random(9)

    def run(groups, W, S, R, max_wg):
        arr = (C.c_uint32 * len(groups))(*groups)
        out = (C.c_uint32 * 2)()
        n = L.sw_hip_test_units(arr, len(groups), W, S, R, max_wg, out)
        assert n > 0, (n, len(groups), W, S, R, max_wg)
        return n, out[0], out[1]

    # the bench shape: 782 groups of 63 strips, 16-wave workgroups on 256 CUs
    n, n_short, short = run([63] * 782, 16, 16, 16 + 256 * 2 + 8, 256)
    assert n >= (782 * 63 + 15) // 16 and n_short > 0 and short <= 4
    # a multiple of the machine: nothing to cut (12 rounds exactly is not reachable with 63-strip groups, use 64)
    n, n_short, _ = run([64] * 768, 16, 16, 600, 256)
    assert n == 768 * 4 and n_short == 0
    # small databases (4-wave workgroups), skewed lengths, single-strip groups, random geometry
    run([63] * 8, 4, 8, 8 + 768 + 8, 768)
    run([1] * 1000, 16, 8, 300, 256)
    run([rng.randint(13, 313) for _ in range(7813)], 16, 64, 64 + 256 * 9 + 8, 256)
    for _ in range(60):
        groups = [rng.randint(1, rng.choice([3, 40, 400])) for _ in range(rng.randint(1, 300))]
        W = rng.choice([4, 8, 16])
        max_wg = rng.choice([3, 64, 256, 512, 768])
        S = rng.randint(1, 70)
        run(groups, W, S, S + max_wg * rng.randint(2, 9) + 8, max_wg)


def test_no_exception_crosses_the_c_abi(L):
    """The reference turns every internal failure into NULL / an error code (rust/src/bindings.rs:109-112) and a cgo
    caller cannot unwind a C++ exception: an allocation failure inside an entry point must come back as an error
    code with a message.  Provoked in a child process under an address-space limit (host arithmetic only, no GPU)."""
    import subprocess
    import sys
    code = r'''
import ctypes, resource
L = ctypes.CDLL(%r)
L.sw_hip_test_units.restype = ctypes.c_int64
L.sw_hip_test_units.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32] + [ctypes.c_uint32] * 4 + [ctypes.POINTER(ctypes.c_uint32)]
L.sw_hip_last_error.restype = ctypes.c_char_p
resource.setrlimit(resource.RLIMIT_AS, (3 << 30, 3 << 30))
g = (ctypes.c_uint32 * 4)(*[1 << 30] * 4)
out = (ctypes.c_uint32 * 2)()
rc = L.sw_hip_test_units(g, 4, 16, 16, 64, 256, out)
print(rc, L.sw_hip_last_error().decode())
''' % os.path.join(ROOT, "dlsa_amd", "lib", "libsw_hip.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split()[0] == "-101" and "out of host memory" in r.stdout, r.stdout
