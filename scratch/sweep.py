#!/usr/bin/env python3
"""Shape sweep on the GPU box: the other BASELINE.json configurations through the same C ABI."""
import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dlsa_amd
from dlsa_amd import bridge

def synth(rng, lens):
    lut = np.frombuffer(b"ATCG", dtype=np.uint8)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64); offs[1:] = np.cumsum(lens)
    flat = lut[rng.integers(0, 4, size=int(offs[-1]), dtype=np.uint8)]
    return flat, offs

def db_case(name, qlen, lens, reps=2):
    rng = np.random.default_rng(7)
    q = synth(rng, np.array([qlen]))[0].tobytes()
    flat, offs = synth(rng, lens)
    t0 = time.perf_counter(); db = bridge.Database(flat=flat, offsets=offs); tb = time.perf_counter() - t0
    sc = dlsa_amd.AlignmentScore(2, -1, -1)
    best = 1e9
    for _ in range(reps + 1):
        t0 = time.perf_counter(); hits = db.search_topk(q, sc, 5); dt = time.perf_counter() - t0
        best = min(best, dt)
    st = bridge.last_stats()
    cells = float(qlen) * float(lens.sum())
    print(json.dumps({"case": name, "pairs": len(lens), "cells": cells, "db_build_s": round(tb, 3), "search_ms": round(best * 1e3, 3),
                      "gcups": round(cells / best / 1e9, 1), "stats": st, "top": hits[0]["score"] if hits else None}), flush=True)
    db.close()

def db_batch_case(name, qlen, nq, lens, reps=2):
    """nq queries of one length through sw_db_search_topk_batch (consecutive short queries share score passes)"""
    rng = np.random.default_rng(7)
    qs = [synth(rng, np.array([qlen]))[0].tobytes() for _ in range(nq)]
    flat, offs = synth(rng, lens)
    db = bridge.Database(flat=flat, offsets=offs)
    sc = dlsa_amd.AlignmentScore(2, -1, -1)
    best = 1e9
    for _ in range(reps + 1):
        t0 = time.perf_counter(); hits = db.search_topk_batch(qs, sc, 5); dt = time.perf_counter() - t0
        best = min(best, dt)
    st = bridge.last_stats()
    cells = float(qlen) * float(lens.sum()) * nq
    print(json.dumps({"case": name, "queries": nq, "cells": cells, "batch_ms": round(best * 1e3, 3), "ms_per_query": round(best * 1e3 / nq, 3),
                      "gcups": round(cells / best / 1e9, 1), "score_launches": st["score_launches"], "score_kernel_ms": round(st["score_kernel_ms"], 3),
                      "top": [h[0]["score"] if h else None for h in hits][:4]}), flush=True)
    db.close()

def pair_case(name, Q, T, mutate=None):
    rng = np.random.default_rng(11)
    q = synth(rng, np.array([Q]))[0]
    if mutate is None:
        t = synth(rng, np.array([T]))[0]
    else:
        t = q.copy()[:T]; idx = rng.random(T) < mutate; t[idx] = np.frombuffer(b"ATCG", dtype=np.uint8)[rng.integers(0, 4, size=int(idx.sum()))]
    sc = dlsa_amd.AlignmentScore(2, -1, -1)
    best = 1e9
    for _ in range(2):
        k0 = bridge.last_stats()["exact_kernel_ms"]
        t0 = time.perf_counter(); r = bridge.align_batch([q.tobytes()], [t.tobytes()], sc, 0); dt = time.perf_counter() - t0
        best = min(best, dt)
        kms = bridge.last_stats()["exact_kernel_ms"] - k0
    dt = best
    r0 = r[0]
    print(json.dumps({"case": name, "Q": Q, "T": T, "s": round(dt, 4), "gcups": round(Q * T / dt / 1e9, 2), "kernels_ms": round(kms, 3),
                      "score": None if r0 is None else r0.score, "alen": None if r0 is None else len(r0.query)}), flush=True)

def batch_case(name, qlen, lens, nq=1):
    """ExecuteWork's shape: every pair of the package through sw_align_batch (exact kernel + traceback for all)."""
    rng = np.random.default_rng(5)
    qs = [synth(rng, np.array([qlen]))[0].tobytes() for _ in range(nq)]
    flat, offs = synth(rng, lens)
    ts = [flat[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(len(lens))]
    Qs = [q for q in qs for _ in ts]; Ts = ts * nq
    sc = dlsa_amd.AlignmentScore(2, -1, -1)
    best = 1e9
    for _ in range(3):
        k0 = bridge.last_stats()["exact_kernel_ms"]
        import ctypes as C
        L = bridge.lib(); n = len(Qs)
        qa = (C.c_char_p * n)(*Qs); ta = (C.c_char_p * n)(*Ts); out = (C.POINTER(bridge.CResult) * n)()
        t0 = time.perf_counter(); rc = L.sw_align_batch(qa, ta, n, sc.c(), 0, out); dt = time.perf_counter() - t0
        assert rc == 0
        t1 = time.perf_counter()
        for i in range(n): L.free_alignment_result(out[i])
        tfree = time.perf_counter() - t1
        best = min(best, dt); kms = bridge.last_stats()["exact_kernel_ms"] - k0
    cells = float(qlen) * float(lens.sum()) * nq
    print(json.dumps({"case": name, "pairs": len(Qs), "cells": cells, "ms": round(best * 1e3, 2), "gcups": round(cells / best / 1e9, 1),
                      "free_ms": round(tfree * 1e3, 2), "kernels_ms": round(kms, 3), "kernel_gcups": round(cells / kms / 1e6, 1)}), flush=True)

def package_case(name, nq, qmin, qmax, nt, tmin, tmax):
    """ExecuteWork on the reference's own synthetic distribution (utils/generate_synthetic_data.py defaults:
    10 queries of 200-1000 nt, 800 targets of 10-200 kb): every pair of the cross product through sw_align_batch."""
    qf, qo = bridge.synth(101, nq, qmin, qmax)
    tf, to = bridge.synth(102, nt, tmin, tmax)
    qs = [qf[int(qo[i]):int(qo[i + 1])].tobytes() for i in range(nq)]
    ts = [tf[int(to[i]):int(to[i + 1])].tobytes() for i in range(nt)]
    import ctypes as C
    L = bridge.lib()
    qb = [C.create_string_buffer(x) for x in qs]; tb = [C.create_string_buffer(x) for x in ts]
    n = nq * nt
    qa = (C.c_char_p * n)(*[C.cast(qb[i // nt], C.c_char_p) for i in range(n)])
    ta = (C.c_char_p * n)(*[C.cast(tb[i % nt], C.c_char_p) for i in range(n)])
    sc = dlsa_amd.AlignmentScore(2, -1, -1)
    cells = float(sum(len(q) for q in qs)) * float(sum(len(t) for t in ts))
    best = 1e9
    for _ in range(2):
        out = (C.POINTER(bridge.CResult) * n)()
        t0 = time.perf_counter(); rc = L.sw_align_batch(qa, ta, n, sc.c(), 0, out); dt = time.perf_counter() - t0
        assert rc == 0, bridge.last_error()
        chk = sum(int(out[i].contents.score) for i in range(0, n, max(1, n // 50)))
        for i in range(n): L.free_alignment_result(out[i])
        best = min(best, dt)
    print(json.dumps({"case": name, "pairs": n, "cells": cells, "ms": round(best * 1e3, 2), "gcups": round(cells / best / 1e9, 1),
                      "package": bridge.package_stats(), "check": chk}), flush=True)

def work_case(name, qlen, lens, nq=1):
    """sw_execute_work + sw_alignment_details_free for a whole package: what a Go worker calls (one C call each)."""
    import ctypes as C
    rng = np.random.default_rng(5)
    qs = [synth(rng, np.array([qlen]))[0].tobytes() for _ in range(nq)]
    flat, offs = synth(rng, lens)
    ts = [flat[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(len(lens))]
    seqs = qs + ts
    L = bridge.lib()
    arr = (C.c_char_p * len(seqs))(*seqs)
    n = nq * len(ts)
    pairs = (bridge.CWorkPair * n)(*[bridge.CWorkPair(i // len(ts), nq + i % len(ts)) for i in range(n)])
    pkg = bridge.CWorkPackage(arr, len(seqs), pairs, n, 2, 1, 1)
    best = bestf = 1e9
    for _ in range(3):
        out = (bridge.CAlignmentDetail * n)()
        t0 = time.perf_counter(); rc = L.sw_execute_work(C.byref(pkg), out); dt = time.perf_counter() - t0
        assert rc >= 0, bridge.last_error()
        chk = sum(int(out[i].score) for i in range(0, n, max(1, n // 50)))
        t1 = time.perf_counter(); L.sw_alignment_details_free(out, n); tf = time.perf_counter() - t1
        best = min(best, dt); bestf = min(bestf, tf)
    cells = float(qlen) * float(lens.sum()) * nq
    print(json.dumps({"case": name, "pairs": n, "cells": cells, "call_ms": round(best * 1e3, 2), "free_ms": round(bestf * 1e3, 2),
                      "gcups_incl_free": round(cells / (best + bestf) / 1e9, 1), "check": chk}), flush=True)

def work_big_case(name, nq, nt, L):
    """SURVEY.md 8d's scaled C5 end-to-end run: nq x nt reads of L nt, EVERY pair through sw_execute_work (one record with
    both aligned strings per pair, as an unmodified master wants them)."""
    import ctypes as C, resource
    fq, oq = bridge.synth(201, nq, L, L)
    ft, ot = bridge.synth(202, nt, L, L)
    seqs = [fq[int(oq[i]):int(oq[i + 1])].tobytes() for i in range(nq)] + [ft[int(ot[i]):int(ot[i + 1])].tobytes() for i in range(nt)]
    Lb = bridge.lib()
    arr = (C.c_char_p * len(seqs))(*seqs)
    n = nq * nt
    pr = np.empty((n, 2), dtype=np.uint32)
    pr[:, 0] = np.repeat(np.arange(nq, dtype=np.uint32), nt)
    pr[:, 1] = np.tile(np.arange(nq, nq + nt, dtype=np.uint32), nq)
    pairs = pr.ctypes.data_as(C.POINTER(bridge.CWorkPair))
    pkg = bridge.CWorkPackage(arr, len(seqs), pairs, n, 2, 1, 1)
    out = (bridge.CAlignmentDetail * n)()
    best = bestf = 1e9
    for _ in range(2):
        k0 = bridge.last_stats()
        t0 = time.perf_counter(); rc = Lb.sw_execute_work(C.byref(pkg), out); dt = time.perf_counter() - t0
        assert rc >= 0, bridge.last_error()
        st = bridge.last_stats()
        chk = sum(int(out[i].score) for i in range(0, n, max(1, n // 97)))
        t1 = time.perf_counter(); Lb.sw_alignment_details_free(out, n); tf = time.perf_counter() - t1
        best = min(best, dt); bestf = min(bestf, tf)
    cells = float(L) * L * n
    print(json.dumps({"case": name, "pairs": n, "cells": cells, "call_s": round(best, 3), "free_s": round(bestf, 3),
                      "pairs_per_s": round(n / best), "gcups_call": round(cells / best / 1e9, 1),
                      "lanes_kernel_ms_last_part": round(st["lanes_kernel_ms"], 2), "lanes_pairs_last_part": st["lanes_pairs"],
                      "peak_rss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2), "check": chk}), flush=True)


which = sys.argv[1:]
rng = np.random.default_rng(3)
if "c1" in which: db_case("C1 1kb x 1000x[900,1100]", 1000, rng.integers(900, 1101, size=1000))
if "c1big" in which: db_case("1kb x 100000x[900,1100]", 1000, rng.integers(900, 1101, size=100000))
if "c2" in which: db_case("C2 10kb x 100000x1kb", 10000, np.full(100000, 1000))
if "c2x4" in which: db_case("C2x4 10kb x 400000x1kb", 10000, np.full(400000, 1000))
if "c2q" in which: db_case("C2q 2.5kb x 400000x1kb", 2500, np.full(400000, 1000))
if "c2t" in which: db_case("C2t 10kb x 25000x4kb", 10000, np.full(25000, 4000))
for w in which:
    if w.startswith("gen:"):  # gen:Q:N:T  -- one query of Q nt against N targets of T nt
        _, gq, gn, gt = w.split(":")
        db_case(w, int(gq), np.full(int(gn), int(gt)))
if "c3" in which: db_case("C3 (1 of 32 queries) 1kb x 1e6xU[200,5000]", 1000, rng.integers(200, 5001, size=1000000))
if "c5" in which: db_case("C5 (1 of 1000 queries) 150 x 1e7x150", 150, np.full(10000000, 150))
if "c5s" in which: db_case("C5 scaled 150 x 1e6x150", 150, np.full(1000000, 150))
if "c5b" in which: db_batch_case("C5 batch: 16 x 150 nt vs 1e7 x 150", 150, 16, np.full(10000000, 150))
if "c5sb" in which: db_batch_case("C5 scaled batch: 16 x 150 nt vs 1e6 x 150", 150, 16, np.full(1000000, 150))
if "c3b" in which: db_batch_case("C3 batch: 8 x 1 kb vs 1e6 x U[200,5000]", 1000, 8, rng.integers(200, 5001, size=1000000))
if "p20k" in which: pair_case("pair 20k", 20000, 20000)
if "p50k" in which: pair_case("pair 50k", 50000, 50000)
if "c4" in which: pair_case("C4 100k", 100000, 100000)
if "c4m" in which: pair_case("C4 100k 5% mutated", 100000, 100000, 0.05)
if "b1" in which: batch_case("ExecuteWork C1: 1kb x 1000x[900,1100] all pairs", 1000, rng.integers(900, 1101, size=1000))
if "b2" in which: batch_case("ExecuteWork 32 x 1000 pairs of ~1kb", 1000, rng.integers(900, 1101, size=1000), nq=8)
if "b5" in which: batch_case("ExecuteWork 150 x 100000x150", 150, np.full(100000, 150))
if "strips" in which:
    for Q in (256, 1024, 2560, 25600):
        pair_case(f"strip timing Q={Q}", Q, 100000)
if "b3" in which: package_case("ExecuteWork generate_synthetic_data defaults: 10 x [200,1000] vs 800 x [10k,200k]", 10, 200, 1000, 800, 10000, 200000)
if "b3s" in which: package_case("ExecuteWork 10 x [200,1000] vs 100 x [10k,200k]", 10, 200, 1000, 100, 10000, 200000)
if "w2" in which: work_case("sw_execute_work 8 x 1000 pairs of ~1kb", 1000, rng.integers(900, 1101, size=1000), nq=8)
if "w5e" in which: work_big_case("sw_execute_work scaled C5 e2e: 1000 x 10000 reads of 150 nt, every pair", 1000, 10000, 150)
if "w5m" in which: work_big_case("sw_execute_work 100 x 10000 reads of 150 nt, every pair", 100, 10000, 150)
if "w5" in which: work_case("sw_execute_work 150 x 100000x150", 150, np.full(100000, 150))
