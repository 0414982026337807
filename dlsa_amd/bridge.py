"""ctypes binding of libsw_hip.so, shaped like the reference's cgo bridge.

Mirrors internal/worker/worker_rust_bridge.go (FindRustAlignmentSimdLowMem :105-124, FindRustAlignmentSimd
:84-103, FindRustAlignmentSequential :63-82, GoResult :20-26) and findAlignmentWithFallback
(internal/worker/worker_impl.go:170-202) so that the parity tests read like calls from the Go worker.
Go is not available in the build image; INTEGRATION.md shows the two cgo lines a maintainer changes.

Nothing here computes an alignment: if the library or a GPU is missing the calls raise SwHipError.
"""
import ctypes as C
import os
from dataclasses import dataclass

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # SW_HIP_LIB: load another build of the same library (experiments); default is the in-tree build
    return os.environ.get("SW_HIP_LIB") or os.path.join(_HERE, "lib", "libsw_hip.so")


class SwHipError(RuntimeError):
    pass


class CResult(C.Structure):  # include/sw.h struct Result == rust/header/sw.h:4-11
    _fields_ = [("query_ptr", C.c_void_p), ("target_ptr", C.c_void_p), ("score", C.c_uint16),
                ("max_x", C.c_uint64), ("max_y", C.c_uint64)]


class CScores(C.Structure):  # include/sw.h struct AlignmentScores == rust/header/sw.h:13-18
    _fields_ = [("gap", C.c_uint16), ("match", C.c_uint16), ("miss", C.c_uint16)]


class CHit(C.Structure):  # include/sw_hip.h struct sw_hit
    _fields_ = [("target_index", C.c_uint64), ("score", C.c_int32), ("length", C.c_uint32),
                ("max_x", C.c_uint64), ("max_y", C.c_uint64), ("query_aln", C.c_void_p),
                ("target_aln", C.c_void_p)]


class CWorkPair(C.Structure):  # include/sw_worker.h struct sw_work_pair
    _fields_ = [("query", C.c_uint32), ("target", C.c_uint32)]


class CWorkPackage(C.Structure):  # include/sw_worker.h struct sw_work_package
    _fields_ = [("sequences", C.POINTER(C.c_char_p)), ("n_sequences", C.c_uint64),
                ("pairs", C.POINTER(CWorkPair)), ("n_pairs", C.c_uint64),
                ("match_score", C.c_int32), ("mismatch_penalty", C.c_int32), ("gap_penalty", C.c_int32)]


class CAlignmentDetail(C.Structure):  # include/sw_worker.h struct sw_alignment_detail
    _fields_ = [("query", C.c_uint32), ("target", C.c_uint32), ("query_alignment", C.c_void_p),
                ("target_alignment", C.c_void_p), ("length", C.c_int64), ("score", C.c_int64),
                ("max_x", C.c_int64), ("max_y", C.c_int64), ("score32", C.c_int32), ("entry", C.c_int32),
                ("owner_", C.c_void_p)]


class CStats(C.Structure):  # include/sw_hip.h struct sw_kernel_stats
    _fields_ = [("score_kernel_ms", C.c_double), ("exact_kernel_ms", C.c_double), ("score_cells", C.c_uint64),
                ("exact_cells", C.c_uint64), ("score_launches", C.c_uint32), ("rerun_pairs", C.c_uint32),
                ("score_strip_rows", C.c_uint64), ("score_chain_rows", C.c_uint64),
                ("score_packed_per_row", C.c_uint32), ("score_aux_per_2rows", C.c_uint32),
                ("score_workgroups", C.c_uint32), ("pad_", C.c_uint32),
                ("lanes_kernel_ms", C.c_double), ("lanes_cells", C.c_uint64), ("lanes_padded_cells", C.c_uint64),
                ("lanes_pairs", C.c_uint64)]


@dataclass
class AlignmentScore:
    """internal/smithwaterman/smithwaterman.go:13-17; the worker passes penalties NEGATED
    (worker_impl.go:126-127), i.e. mismatch_penalty and gap_penalty are negative numbers here."""
    match_score: int
    mismatch_penalty: int
    gap_penalty: int

    def c(self):
        return CScores(self.gap_penalty & 0xFFFF, self.match_score & 0xFFFF, self.mismatch_penalty & 0xFFFF)


@dataclass
class GoResult:  # worker_rust_bridge.go:20-26
    query: str
    target: str
    score: int   # uint16
    max_x: int   # uint64
    max_y: int   # uint64
    score32: int = 0


_lib = None


def lib():
    """Loads libsw_hip.so and declares every symbol of include/sw.h and include/sw_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise SwHipError(f"{path} is missing: build it with `python -m dlsa_amd.build` (there is no CPU fallback)")
    L = C.CDLL(path)
    RP = C.POINTER(CResult)
    for name in ("find_alignment_low_memory", "find_alignment_simd", "find_alignment_sequential_straight",
                 "find_alignment_sequential"):
        f = getattr(L, name)
        f.argtypes = [C.c_char_p, C.c_char_p, CScores]
        f.restype = RP
    L.find_alignment_parallel.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, CScores]
    L.find_alignment_parallel.restype = RP
    L.free_alignment_result.argtypes = [RP]
    L.free_alignment_result.restype = None
    L.free_c_string.argtypes = [C.c_void_p]
    L.free_c_string.restype = None
    L.sw_hip_init.argtypes = [C.c_int]
    L.sw_hip_init.restype = C.c_int
    L.sw_hip_last_error.restype = C.c_char_p
    L.sw_hip_version.restype = C.c_char_p
    L.sw_result_score32.argtypes = [RP]
    L.sw_result_score32.restype = C.c_int32
    L.sw_align_batch.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_size_t, CScores, C.c_int,
                                 C.POINTER(RP)]
    L.sw_align_batch.restype = C.c_int
    L.sw_db_create.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.c_uint64]
    L.sw_db_create.restype = C.c_void_p
    L.sw_db_load_fasta.argtypes = [C.c_char_p]
    L.sw_db_load_fasta.restype = C.c_void_p
    L.sw_db_save.argtypes = [C.c_void_p, C.c_char_p]
    L.sw_db_save.restype = C.c_int
    L.sw_db_open.argtypes = [C.c_char_p]
    L.sw_db_open.restype = C.c_void_p
    L.sw_node_db_save.argtypes = [C.c_void_p, C.c_char_p]
    L.sw_node_db_save.restype = C.c_int
    L.sw_node_db_open.argtypes = [C.c_void_p, C.c_char_p]
    L.sw_node_db_open.restype = C.c_void_p
    L.sw_db_free.argtypes = [C.c_void_p]
    L.sw_db_free.restype = None
    L.sw_db_size.argtypes = [C.c_void_p]
    L.sw_db_size.restype = C.c_uint64
    L.sw_db_residues.argtypes = [C.c_void_p]
    L.sw_db_residues.restype = C.c_uint64
    L.sw_db_search_topk.argtypes = [C.c_void_p, C.c_char_p, CScores, C.c_int, C.POINTER(CHit), C.POINTER(C.c_int)]
    L.sw_db_search_topk.restype = C.c_int
    L.sw_db_search_topk_batch.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t, CScores, C.c_int,
                                          C.POINTER(CHit), C.POINTER(C.c_int)]
    L.sw_db_search_topk_batch.restype = C.c_int
    L.sw_db_scores.argtypes = [C.c_void_p, C.c_char_p, CScores, C.POINTER(C.c_int32)]
    L.sw_db_scores.restype = C.c_int
    L.sw_db_scores_noalloc.argtypes = [C.c_void_p, C.c_char_p, CScores, C.POINTER(C.c_int32)]
    L.sw_db_scores_noalloc.restype = C.c_int
    L.sw_hip_noalloc_scratch_bytes.restype = C.c_uint64
    L.sw_hits_free.argtypes = [C.POINTER(CHit), C.c_int]
    L.sw_hits_free.restype = None
    L.sw_hip_last_stats.argtypes = [C.POINTER(CStats)]
    L.sw_hip_last_stats.restype = C.c_int
    L.sw_hip_benchmark_cups.restype = C.c_float
    L.sw_hip_struct_size.argtypes = [C.c_int]
    L.sw_hip_struct_size.restype = C.c_uint32
    for which, mirror in enumerate((CStats, CHit, CAlignmentDetail, CWorkPackage, CWorkPair)):
        if L.sw_hip_struct_size(which) != C.sizeof(mirror):
            raise SwHipError(f"{mirror.__name__} is {C.sizeof(mirror)} bytes, the library's struct "
                             f"{L.sw_hip_struct_size(which)}: dlsa_amd/bridge.py has fallen behind include/*.h")
    L.sw_hip_package_stats.argtypes = [C.POINTER(C.c_uint64)]
    L.sw_hip_package_stats.restype = None
    L.sw_hip_set_scratch_mb.argtypes = [C.c_int, C.c_uint64]
    L.sw_hip_set_scratch_mb.restype = C.c_int
    L.sw_hip_path_params.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
    L.sw_hip_path_params.restype = C.c_int
    L.sw_hip_test_units.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.POINTER(C.c_uint32)]
    L.sw_hip_test_units.restype = C.c_int64
    # one process, several GPUs
    L.sw_node_create.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.sw_node_create.restype = C.c_void_p
    L.sw_node_free.argtypes = [C.c_void_p]
    L.sw_node_free.restype = None
    L.sw_node_shards.argtypes = [C.c_void_p]
    L.sw_node_shards.restype = C.c_int
    L.sw_shard_plan.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]
    L.sw_shard_plan.restype = C.c_int
    L.sw_node_db_create.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint64]
    L.sw_node_db_create.restype = C.c_void_p
    L.sw_node_db_free.argtypes = [C.c_void_p]
    L.sw_node_db_free.restype = None
    L.sw_node_db_size.argtypes = [C.c_void_p]
    L.sw_node_db_size.restype = C.c_uint64
    L.sw_node_db_shards.argtypes = [C.c_void_p]
    L.sw_node_db_shards.restype = C.c_int
    L.sw_node_db_shard_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sw_node_db_shard_info.restype = C.c_int
    L.sw_node_search_topk_batch.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t, CScores, C.c_int,
                                            C.POINTER(CHit), C.POINTER(C.c_int)]
    L.sw_node_search_topk_batch.restype = C.c_int
    L.sw_node_last_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(CStats)]
    L.sw_node_last_stats.restype = C.c_int
    L.sw_node_benchmark_cups.argtypes = [C.c_void_p]
    L.sw_node_benchmark_cups.restype = C.c_float
    # include/sw_worker.h
    L.sw_execute_work.argtypes = [C.POINTER(CWorkPackage), C.POINTER(CAlignmentDetail)]
    L.sw_execute_work.restype = C.c_int
    L.sw_free_alignment_results.argtypes = [C.POINTER(C.POINTER(CResult)), C.c_size_t]
    L.sw_free_alignment_results.restype = None
    L.sw_node_execute_work.argtypes = [C.c_void_p, C.POINTER(CWorkPackage), C.POINTER(CAlignmentDetail)]
    L.sw_node_execute_work.restype = C.c_int
    L.sw_node_align_batch.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_size_t, CScores,
                                      C.c_int, C.POINTER(C.POINTER(CResult))]
    L.sw_node_align_batch.restype = C.c_int
    L.sw_alignment_details_free.argtypes = [C.POINTER(CAlignmentDetail), C.c_uint64]
    L.sw_alignment_details_free.restype = None
    L.sw_benchmark_rust.argtypes = [C.c_double, C.c_int, C.c_int]
    L.sw_benchmark_rust.restype = C.c_float
    L.sw_fasta_parse.argtypes = [C.c_char_p]
    L.sw_fasta_parse.restype = C.c_void_p
    L.sw_fasta_count.argtypes = [C.c_void_p]
    L.sw_fasta_count.restype = C.c_uint64
    L.sw_fasta_id.argtypes = [C.c_void_p, C.c_uint64]
    L.sw_fasta_id.restype = C.c_char_p
    L.sw_fasta_sequence.argtypes = [C.c_void_p, C.c_uint64]
    L.sw_fasta_sequence.restype = C.c_char_p
    L.sw_fasta_free.argtypes = [C.c_void_p]
    L.sw_fasta_free.restype = None
    L.sw_db_from_fasta.argtypes = [C.c_void_p]
    L.sw_db_from_fasta.restype = C.c_void_p
    L.sw_write_query_results.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(CHit), C.c_int]
    L.sw_write_query_results.restype = C.c_int
    L.sw_synth_lengths.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.sw_synth_lengths.restype = C.c_int
    L.sw_synth_fill.argtypes = [C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                C.POINTER(C.c_uint64)]
    L.sw_synth_fill.restype = C.c_int
    L.sw_synth_write_fasta.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    L.sw_synth_write_fasta.restype = C.c_int
    _lib = L
    return L


def last_error():
    return lib().sw_hip_last_error().decode("utf-8", "replace")


def _b(s):
    return s if isinstance(s, bytes) else s.encode("utf-8")


def _convert(L, res):
    """convertResultToGoResult (worker_rust_bridge.go:131-139) + FreeAlignmentResult (:142-144)."""
    r = res.contents
    out = GoResult(C.string_at(r.query_ptr).decode("utf-8"), C.string_at(r.target_ptr).decode("utf-8"),
                   int(r.score), int(r.max_x), int(r.max_y), int(L.sw_result_score32(res)))
    L.free_alignment_result(res)
    return out


def _call(name, msg, query, target, score):
    L = lib()
    res = getattr(L, name)(_b(query), _b(target), score.c())
    if not res:
        return None, SwHipError(f"{msg}: {last_error()}")
    return _convert(L, res), None


def find_rust_alignment_simd_low_mem(query, target, score):
    return _call("find_alignment_low_memory", "Crash in SIMD ringbuffer alignment", query, target, score)


def find_rust_alignment_simd(query, target, score):
    return _call("find_alignment_simd", "Crash in rust SIMD alignment", query, target, score)


def find_rust_alignment_sequential(query, target, score):
    return _call("find_alignment_sequential_straight", "Crash in rust sequential alignment", query, target, score)


def find_alignment_with_fallback(query, target, score):
    """internal/worker/worker_impl.go:170-202: low memory, then simd, then sequential."""
    for f in (find_rust_alignment_simd_low_mem, find_rust_alignment_simd, find_rust_alignment_sequential):
        res, err = f(query, target, score)
        if err is None:
            return res, None
    return None, SwHipError("All alignment functions failed to align a sequence")


def align_batch(queries, targets, score, path=0, node=None):
    """sw_align_batch (node: sw_node_align_batch over that Node's GPUs): one submission for a list of pairs; returns
    a list of GoResult/None."""
    L = lib()
    n = len(queries)
    # one C string per distinct Python string: a work package names the same sequence over and over
    # (worker_impl.go:115-117), and the library groups a package by the pointers it is given
    enc = {}
    def cstr(s):
        b = enc.get(s)
        if b is None:
            b = enc[s] = C.create_string_buffer(_b(s))
        return C.cast(b, C.c_char_p)
    qs = (C.c_char_p * n)(*[cstr(q) for q in queries])
    ts = (C.c_char_p * n)(*[cstr(t) for t in targets])
    out = (C.POINTER(CResult) * n)()
    if node is not None:
        rc = L.sw_node_align_batch(node._h, qs, ts, n, score.c(), path, out)
    else:
        rc = L.sw_align_batch(qs, ts, n, score.c(), path, out)
    if rc != 0:
        raise SwHipError(f"sw_align_batch failed ({rc}): {last_error()}")
    return [(_convert(L, out[i]) if out[i] else None) for i in range(n)]


class Database:
    """A resident target database (sw_db_*)."""

    def __init__(self, seqs=None, flat=None, offsets=None, fasta=None, path=None):
        L = lib()
        if path is not None:  # a file written by save()
            self._h = L.sw_db_open(_b(path))
            if not self._h:
                raise SwHipError(f"sw_db_open failed: {last_error()}")
            self.n = int(L.sw_db_size(self._h))
            return
        if fasta is not None:  # a Fasta object (dlsa_amd.worker)
            self._h = L.sw_db_from_fasta(fasta._h)
            n = len(fasta)
        elif seqs is not None:
            bs = [_b(s) for s in seqs]
            flat = b"".join(bs)
            offs = [0]
            for s in bs:
                offs.append(offs[-1] + len(s))
            n = len(bs)
            off_arr = (C.c_uint64 * (n + 1))(*offs)
            self._h = L.sw_db_create(flat, off_arr, n)
        else:
            import numpy as np
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n = len(offsets) - 1
            buf = flat if isinstance(flat, bytes) else flat.tobytes()
            self._h = L.sw_db_create(buf, offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        if not self._h:
            raise SwHipError(f"sw_db_create failed: {last_error()}")
        self.n = n

    def save(self, path):
        rc = lib().sw_db_save(self._h, _b(path))
        if rc != 0:
            raise SwHipError(f"sw_db_save failed ({rc}): {last_error()}")

    def close(self):
        if self._h:
            lib().sw_db_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scores(self, query, score):
        import numpy as np
        out = np.zeros(self.n, dtype=np.int32)
        rc = lib().sw_db_scores(self._h, _b(query), score.c(), out.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc != 0:
            raise SwHipError(f"sw_db_scores failed ({rc}): {last_error()}")
        return out

    def scores_noalloc(self, query, score):
        """sw_db_scores_noalloc: the same scores without any device allocation (fixed scratch, stated limits)."""
        import numpy as np
        out = np.zeros(self.n, dtype=np.int32)
        rc = lib().sw_db_scores_noalloc(self._h, _b(query), score.c(), out.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc != 0:
            raise SwHipError(f"sw_db_scores_noalloc failed ({rc}): {last_error()}")
        return out

    @staticmethod
    def _hit(h):
        return dict(target_index=int(h.target_index), score=int(h.score), length=int(h.length),
                    max_x=int(h.max_x), max_y=int(h.max_y),
                    query_aln=C.string_at(h.query_aln).decode(), target_aln=C.string_at(h.target_aln).decode())

    def search_topk(self, query, score, k):
        L = lib()
        hits = (CHit * max(k, 1))()
        nh = C.c_int(0)
        rc = L.sw_db_search_topk(self._h, _b(query), score.c(), k, hits, C.byref(nh))
        if rc != 0:
            raise SwHipError(f"sw_db_search_topk failed ({rc}): {last_error()}")
        out = [self._hit(hits[i]) for i in range(nh.value)]
        L.sw_hits_free(hits, nh.value)
        return out

    def search_topk_batch(self, queries, score, k):
        """sw_db_search_topk_batch: one list of hits per query."""
        L = lib()
        n = len(queries)
        qs = (C.c_char_p * max(n, 1))(*[_b(q) for q in queries])
        hits = (CHit * max(n * k, 1))()
        nh = (C.c_int * max(n, 1))()
        rc = L.sw_db_search_topk_batch(self._h, qs, n, score.c(), k, hits, nh)
        if rc != 0:
            raise SwHipError(f"sw_db_search_topk_batch failed ({rc}): {last_error()}")
        out = []
        for q in range(n):
            out.append([self._hit(hits[q * k + i]) for i in range(nh[q])])
            L.sw_hits_free(C.cast(C.byref(hits, q * k * C.sizeof(CHit)), C.POINTER(CHit)), nh[q])
        return out


def synth_lengths(seed, n, min_len, max_len, first=0):
    """sw_synth_lengths: lengths of sequences first .. first + n - 1 of the seeded synthetic set."""
    import numpy as np
    out = np.zeros(max(n, 1), dtype=np.uint32)
    rc = lib().sw_synth_lengths(seed, first, n, min_len, max_len, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc != 0:
        raise SwHipError(f"sw_synth_lengths failed ({rc}): {last_error()}")
    return out[:n]


def synth(seed, n, min_len, max_len, indices=None):
    """The seeded restatement of utils/generate_synthetic_data.py:7-12 (sw_synth_*): returns (flat uint8 array,
    offsets uint64 array) of sequences `indices` (default 0 .. n-1) of the set (seed, min_len, max_len)."""
    import numpy as np
    L = lib()
    if indices is None:
        lens = synth_lengths(seed, n, min_len, max_len)
        idx_p = None
    else:
        indices = np.ascontiguousarray(indices, dtype=np.uint64)
        n = len(indices)
        if min_len == max_len:
            lens = np.full(n, min_len, dtype=np.uint32)
        else:  # lengths of arbitrary indices: one call per contiguous run would do; simply generate them all once
            hi = int(indices.max()) + 1 if n else 0
            lens = synth_lengths(seed, hi, min_len, max_len)[indices.astype(np.int64)]
        idx_p = indices.ctypes.data_as(C.POINTER(C.c_uint64))
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens, dtype=np.uint64)
    flat = np.empty(max(int(offsets[-1]), 1), dtype=np.uint8)
    rc = L.sw_synth_fill(seed, idx_p, n, min_len, max_len, flat.ctypes.data_as(C.c_void_p),
                         offsets.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc != 0:
        raise SwHipError(f"sw_synth_fill failed ({rc}): {last_error()}")
    return flat[:int(offsets[-1])], offsets


def shard_plan(offsets, n_shards):
    """sw_shard_plan: shard of every sequence of a database given by its n + 1 offsets (host arithmetic, no GPU)."""
    import numpy as np
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    out = np.zeros(max(n, 1), dtype=np.uint32)
    rc = lib().sw_shard_plan(offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n, n_shards,
                             out.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc != 0:
        raise SwHipError(f"sw_shard_plan failed ({rc}): {last_error()}")
    return out[:n]


class Node:
    """sw_node: the GPUs one worker process drives (devices may repeat: logical shards on one GPU)."""

    def __init__(self, devices=None, n_devices=0):
        L = lib()
        if devices is None:
            self._h = L.sw_node_create(None, n_devices)
        else:
            arr = (C.c_int * len(devices))(*devices)
            self._h = L.sw_node_create(arr, len(devices))
        if not self._h:
            raise SwHipError(f"sw_node_create failed: {last_error()}")
        self.n_shards = L.sw_node_shards(self._h)

    def benchmark_cups(self):
        return float(lib().sw_node_benchmark_cups(self._h))

    def close(self):
        if self._h:
            lib().sw_node_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NodeDatabase:
    """sw_node_db: one database sharded by target over the GPUs of a Node."""

    def __init__(self, node, seqs=None, flat=None, offsets=None, path=None):
        import numpy as np
        L = lib()
        self._node = node
        if path is not None:  # a directory written by save()
            self._h = L.sw_node_db_open(node._h, _b(path))
            if not self._h:
                raise SwHipError(f"sw_node_db_open failed: {last_error()}")
            self.n = int(L.sw_node_db_size(self._h))
            return
        if seqs is not None:
            bs = [_b(s) for s in seqs]
            flat = b"".join(bs)
            offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
            offsets[1:] = np.cumsum([len(s) for s in bs], dtype=np.uint64)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.n = len(offsets) - 1
        buf = flat if isinstance(flat, bytes) else flat.tobytes()
        self._node = node
        self._h = L.sw_node_db_create(node._h, buf, offsets.ctypes.data_as(C.POINTER(C.c_uint64)), self.n)
        if not self._h:
            raise SwHipError(f"sw_node_db_create failed: {last_error()}")

    def save(self, path):
        rc = lib().sw_node_db_save(self._h, _b(path))
        if rc != 0:
            raise SwHipError(f"sw_node_db_save failed ({rc}): {last_error()}")

    def shard_info(self):
        L = lib()
        out = []
        for s in range(L.sw_node_db_shards(self._h)):
            dev, n, res = C.c_int(), C.c_uint64(), C.c_uint64()
            L.sw_node_db_shard_info(self._h, s, C.byref(dev), C.byref(n), C.byref(res))
            out.append(dict(device=dev.value, n_targets=n.value, residues=res.value))
        return out

    def search_topk_batch(self, queries, score, k):
        L = lib()
        n = len(queries)
        qs = (C.c_char_p * max(n, 1))(*[_b(q) for q in queries])
        hits = (CHit * max(n * k, 1))()
        nh = (C.c_int * max(n, 1))()
        rc = L.sw_node_search_topk_batch(self._h, qs, n, score.c(), k, hits, nh)
        if rc != 0:
            raise SwHipError(f"sw_node_search_topk_batch failed ({rc}): {last_error()}")
        out = []
        for q in range(n):
            out.append([Database._hit(hits[q * k + i]) for i in range(nh[q])])
            L.sw_hits_free(C.cast(C.byref(hits, q * k * C.sizeof(CHit)), C.POINTER(CHit)), nh[q])
        return out

    def last_stats(self, shard):
        st = CStats()
        lib().sw_node_last_stats(self._h, shard, C.byref(st))
        return {k: getattr(st, k) for k, _ in CStats._fields_}

    def close(self):
        if self._h:
            lib().sw_node_db_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def package_stats():
    """sw_hip_package_stats: what the work-package fast path did in the last batch call."""
    out = (C.c_uint64 * 7)()
    lib().sw_hip_package_stats(out)
    return dict(zip(("pairs", "queries", "windows", "rounds", "window_cells", "chunks", "chunk_escapes"), [int(x) for x in out]))


def set_scratch_mb(mb, device=-1):
    """sw_hip_set_scratch_mb: the device's scratch budget (exact submissions are cut, work packages sliced by it)."""
    rc = lib().sw_hip_set_scratch_mb(device, mb)
    if rc != 0:
        raise SwHipError(f"sw_hip_set_scratch_mb failed ({rc}): {last_error()}")


def last_stats():
    st = CStats()
    lib().sw_hip_last_stats(C.byref(st))
    return {k: getattr(st, k) for k, _ in CStats._fields_}
