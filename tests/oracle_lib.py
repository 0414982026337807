"""ctypes binding of oracle/liboracle.so (the CPU restatement).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by dlsa_amd/."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "liboracle.so")

PATH_LOWMEM, PATH_SIMD, PATH_STRAIGHT = 0, 1, 2
OK, PANIC, NOMEM = 0, 1, 2
U64MAX = (1 << 64) - 1


class Scores(C.Structure):
    _fields_ = [("gap", C.c_int16), ("match", C.c_int16), ("miss", C.c_int16)]


class _Result(C.Structure):
    _fields_ = [("q_aln", C.POINTER(C.c_uint32)), ("t_aln", C.POINTER(C.c_uint32)), ("aln_len", C.c_size_t),
                ("score", C.c_int16), ("max_x", C.c_uint64), ("max_y", C.c_uint64)]


class _CleanResult(C.Structure):
    _fields_ = [("q_aln", C.POINTER(C.c_uint32)), ("t_aln", C.POINTER(C.c_uint32)), ("aln_len", C.c_size_t),
                ("score32", C.c_int32), ("max_x", C.c_uint64), ("max_y", C.c_uint64)]


def _needs_build():
    if not os.path.exists(_SO):
        return True
    m = os.path.getmtime(_SO)
    return any(os.path.getmtime(os.path.join(ORACLE_DIR, f)) > m
               for f in ("sw_oracle.c", "sw_clean.c", "sw_oracle.h", "Makefile"))


def build():
    if _needs_build():
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        sig = [C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t, Scores, C.POINTER(_Result)]
        for name in ("oracle_find_alignment_simd", "oracle_find_alignment_simd_lowmem",
                     "oracle_find_alignment_sequential_straight", "oracle_find_alignment_sequential"):
            getattr(L, name).argtypes = sig
            getattr(L, name).restype = C.c_int
        L.oracle_find_alignment_with_fallback.argtypes = sig + [C.POINTER(C.c_int)]
        L.oracle_find_alignment_with_fallback.restype = C.c_int
        L.oracle_result_free.argtypes = [C.POINTER(_Result)]
        L.oracle_clean_align.argtypes = sig[:5] + [C.c_int, C.POINTER(_CleanResult)]
        L.oracle_clean_align.restype = C.c_int
        L.oracle_clean_result_free.argtypes = [C.POINTER(_CleanResult)]
        L.oracle_clean_score_plain.argtypes = sig[:5]
        L.oracle_clean_score_plain.restype = C.c_int32
        L.oracle_bench_pairs.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_char_p),
                                         C.POINTER(C.c_size_t), C.c_size_t, Scores, C.c_int,
                                         C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.oracle_bench_pairs.restype = C.c_double
        _lib = L
    return _lib


def _cps(s):
    if isinstance(s, bytes):
        s = s.decode("utf-8")
    arr = (C.c_uint32 * max(1, len(s)))(*[ord(c) for c in s])
    return arr, len(s)


def _str(ptr, n):
    return "".join(chr(ptr[i]) for i in range(n))


FUNCS = {
    "simd": "oracle_find_alignment_simd",
    "lowmem": "oracle_find_alignment_simd_lowmem",
    "straight": "oracle_find_alignment_sequential_straight",
    "sequential": "oracle_find_alignment_sequential",
}


def align(func, query, target, gap, match, miss):
    """Returns dict(q, t, score, max_x, max_y) or None where the reference would panic (-> NULL)."""
    L = lib()
    qa, Q = _cps(query)
    ta, T = _cps(target)
    r = _Result()
    rc = getattr(L, FUNCS[func])(qa, Q, ta, T, Scores(gap, match, miss), C.byref(r))
    if rc != OK:
        return None
    out = dict(q=_str(r.q_aln, r.aln_len), t=_str(r.t_aln, r.aln_len), score=int(r.score),
               max_x=int(r.max_x), max_y=int(r.max_y))
    L.oracle_result_free(C.byref(r))
    return out


def align_with_fallback(query, target, gap, match, miss):
    L = lib()
    qa, Q = _cps(query)
    ta, T = _cps(target)
    r = _Result()
    which = C.c_int(-1)
    rc = L.oracle_find_alignment_with_fallback(qa, Q, ta, T, Scores(gap, match, miss), C.byref(r), C.byref(which))
    if rc != OK:
        return None
    out = dict(q=_str(r.q_aln, r.aln_len), t=_str(r.t_aln, r.aln_len), score=int(r.score),
               max_x=int(r.max_x), max_y=int(r.max_y), which=which.value)
    L.oracle_result_free(C.byref(r))
    return out


def clean_align(path, query, target, gap, match, miss):
    L = lib()
    qa, Q = _cps(query)
    ta, T = _cps(target)
    r = _CleanResult()
    rc = L.oracle_clean_align(qa, Q, ta, T, Scores(gap, match, miss), path, C.byref(r))
    if rc != OK:
        return None
    out = dict(q=_str(r.q_aln, r.aln_len), t=_str(r.t_aln, r.aln_len), score=int(r.score32),
               max_x=int(r.max_x), max_y=int(r.max_y))
    L.oracle_clean_result_free(C.byref(r))
    return out


def clean_score_plain(query, target, gap, match, miss):
    qa, Q = _cps(query)
    ta, T = _cps(target)
    return int(lib().oracle_clean_score_plain(qa, Q, ta, T, Scores(gap, match, miss)))


def bench_pairs(queries, targets, gap, match, miss, threads):
    """queries/targets: lists of bytes.  Returns (seconds, cells, scores)."""
    n = len(queries)
    qs = (C.c_char_p * n)(*queries)
    ts = (C.c_char_p * n)(*targets)
    ql = (C.c_size_t * n)(*[len(q) for q in queries])
    tl = (C.c_size_t * n)(*[len(t) for t in targets])
    cells = C.c_double(0)
    sc = (C.c_int32 * n)()
    secs = lib().oracle_bench_pairs(qs, ql, ts, tl, n, Scores(gap, match, miss), threads, C.byref(cells), sc)
    return secs, cells.value, list(sc)
