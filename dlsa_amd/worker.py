"""The worker-side callers of the hot path, shaped like the reference's Go worker and Python CLI.

Mirrors (over the C entry points of include/sw_worker.h; the logic itself lives in dlsa_amd/csrc/sw_worker.cpp):
  * the wire structs ExecuteWork fills -- QueryTargetType, Alignment, AlignmentDetail, WorkResult, WorkPackage,
    CompleteWorkPackage (internal/worker/master_rest_client.go:16-68) -- with the reference's JSON field names;
  * Worker.ExecuteWork / ExecuteWorkInParallel (internal/worker/worker_impl.go:108-168, 204-238);
  * BenchmarkRust (internal/worker/worker_rust_benchmark.go:38-68) and the GPU capacity figure a batch worker
    registers (POST /worker/register {"benchmark_result": ...}, master_rest_client.go:44-46, 82-101);
  * parse_fasta and the result files of the CLI (cli/__main__.py:19-34, 153-194).

Go is not available where this is built; these classes exist so that the tests read like the reference's
callers and so that a Python driver can run a work package end to end.  The REST client, heartbeat and
registration (master_rest_client.go:70-204) are out of scope.  Nothing here computes an alignment.
"""
import ctypes as C
import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

from . import bridge
from .bridge import SwHipError

BUFFER_SIZE = 100  # worker_impl.go:112


@dataclass
class QueryTargetType:  # master_rest_client.go:16-19
    query: str
    target: str


@dataclass
class WorkPackage:  # master_rest_client.go:30-37 (penalties are positive numbers on the wire)
    id: Optional[str]
    job_id: Optional[str]
    queries: List[QueryTargetType]
    match_score: int
    mismatch_penalty: int
    gap_penalty: int

    @classmethod
    def from_json(cls, text):
        d = json.loads(text) if isinstance(text, (str, bytes)) else text
        return cls(d.get("id"), d.get("job_id"), [QueryTargetType(q["query"], q["target"]) for q in d["queries"]],
                   int(d["match_score"]), int(d["mismatch_penalty"]), int(d["gap_penalty"]))


@dataclass
class CompleteWorkPackage(WorkPackage):  # master_rest_client.go:39-42
    sequences: Dict[str, str] = field(default_factory=dict)


@dataclass
class Alignment:  # master_rest_client.go:21-28
    query_alignment: str
    target_alignment: str
    length: int
    score: int
    maxX: int
    maxY: int


@dataclass
class AlignmentDetail:  # master_rest_client.go:65-68
    combination: Dict[str, str]  # TargetQueryCombination: {"target": id, "query": id} (:56-59)
    alignment: Alignment
    score32: int = 0   # additive, not on the wire
    entry: int = 0     # additive: which entry point of the fallback chain answered

    def to_wire(self):
        a = self.alignment
        return {"combination": {"target": self.combination["target"], "query": self.combination["query"]},
                "alignment": {"query_alignment": a.query_alignment, "target_alignment": a.target_alignment,
                              "length": a.length, "score": a.score, "maxX": a.maxX, "maxY": a.maxY}}


@dataclass
class WorkResult:  # master_rest_client.go:61-63
    alignments: List[AlignmentDetail] = field(default_factory=list)

    def to_json(self):
        return json.dumps({"alignments": [a.to_wire() for a in self.alignments]})


class Worker:
    """Worker of internal/worker/worker_impl.go:18-23 without the REST client: results are handed to `send_result`
    (RestClient.SendResult, :161,166) -- by default collected in self.sent."""

    def __init__(self, send_result=None, node=None):
        """node: a bridge.Node -- every work package is then spread over that node's GPUs (sw_node_execute_work)."""
        self.sent: List[WorkResult] = []
        self._send = send_result or self.sent.append
        self.node = node

    def align(self, work: CompleteWorkPackage, queries: List[QueryTargetType]) -> List[AlignmentDetail]:
        """All AlignmentDetails of `queries`, in order: one sw_execute_work call."""
        L = bridge.lib()
        ids = list(work.sequences.keys())
        index = {sid: i for i, sid in enumerate(ids)}
        for comb in queries:  # worker_impl.go:116-122
            if comb.query not in index or comb.target not in index:
                raise SwHipError("Target and query not found in work package")
        n = len(queries)
        if n == 0:
            return []
        seqs = (C.c_char_p * len(ids))(*[work.sequences[s].encode("utf-8") for s in ids])
        pairs = (bridge.CWorkPair * n)(*[bridge.CWorkPair(index[c.query], index[c.target]) for c in queries])
        pkg = bridge.CWorkPackage(seqs, len(ids), pairs, n, work.match_score, work.mismatch_penalty, work.gap_penalty)
        out = (bridge.CAlignmentDetail * n)()
        if self.node is not None:
            rc = L.sw_node_execute_work(self.node._h, C.byref(pkg), out)
        else:
            rc = L.sw_execute_work(C.byref(pkg), out)
        if rc < 0:  # (rc > 0: that many pairs failed on every entry point -- entry == -1 below)
            raise SwHipError(f"sw_execute_work failed ({rc}): {bridge.last_error()}")
        self.failed_pairs = rc
        res = []
        for i, comb in enumerate(queries):
            d = out[i]
            res.append(AlignmentDetail(
                {"target": comb.target, "query": comb.query},
                Alignment(C.string_at(d.query_alignment).decode("utf-8"), C.string_at(d.target_alignment).decode("utf-8"),
                          int(d.length), int(d.score), int(d.max_x), int(d.max_y)),
                int(d.score32), int(d.entry)))
        L.sw_alignment_details_free(out, n)
        return res

    def execute_work(self, work: CompleteWorkPackage, queries: List[QueryTargetType]):
        """ExecuteWork (worker_impl.go:108-168): same results, same batching of the posts -- a buffer is sent when
        idx % 100 == 0 and idx > 0 (so the first post holds 101 results), the rest at the end."""
        details = self.align(work, queries)
        buf = WorkResult()
        for idx, d in enumerate(details):
            buf.alignments.append(d)
            if idx % BUFFER_SIZE == 0 and idx > 0:
                self._send(buf)
                buf = WorkResult()
        self._send(buf)

    def execute_work_in_parallel(self, work: CompleteWorkPackage, n_cpus=0):
        """ExecuteWorkInParallel (worker_impl.go:204-238) splits work.queries into one contiguous chunk per CPU
        and runs ExecuteWork on each.  A GPU is fed best with the whole package at once, so the chunks only
        decide how the results are grouped into posts: chunk i is [n*i/c, n*(i+1)/c) as in the reference."""
        cpu_count = (os.cpu_count() or 2) - 1
        if n_cpus != 0 and n_cpus - 1 < cpu_count:
            cpu_count = n_cpus
        cpu_count = max(cpu_count, 1)
        n = len(work.queries)
        details = self.align(work, work.queries)
        for i in range(cpu_count):
            chunk = details[n * i // cpu_count: n * (i + 1) // cpu_count]
            buf = WorkResult()
            for idx, d in enumerate(chunk):
                buf.alignments.append(d)
                if idx % BUFFER_SIZE == 0 and idx > 0:
                    self._send(buf)
                    buf = WorkResult()
            self._send(buf)


def benchmark_rust(threshold_seconds=0.010, q_steps=1, t_steps=2):
    """BenchmarkRust(10 ms, 1, 2) of cmd/worker/main.go:43 on this library: cells per second of ONE pair."""
    v = float(bridge.lib().sw_benchmark_rust(threshold_seconds, q_steps, t_steps))
    if v <= 0:
        raise SwHipError(f"sw_benchmark_rust failed: {bridge.last_error()}")
    return v


def benchmark_capacity():
    """The number a batch GPU worker registers: cells per second of the database path, integer-valued
    (the master's WorkerResources.benchmark_result is a pydantic int, master/api_models/worker.py:7-16)."""
    v = float(bridge.lib().sw_hip_benchmark_cups())
    if v <= 0:
        raise SwHipError(f"sw_hip_benchmark_cups failed: {bridge.last_error()}")
    return v


def machine_specs_request(benchmark):
    """MachineSpecsRequest (master_rest_client.go:44-46): the JSON body of POST /worker/register."""
    return json.dumps({"benchmark_result": benchmark})


class Fasta:
    """parse_fasta of cli/__main__.py:19-34 (sw_fasta_parse): (id, sequence) records."""

    def __init__(self, path):
        L = bridge.lib()
        self._h = L.sw_fasta_parse(os.fsencode(path))
        if not self._h:
            raise SwHipError(f"sw_fasta_parse failed: {bridge.last_error()}")
        n = int(L.sw_fasta_count(self._h))
        self.ids = [L.sw_fasta_id(self._h, i).decode("utf-8") for i in range(n)]

    def __len__(self):
        return len(self.ids)

    def sequence(self, i):
        return bridge.lib().sw_fasta_sequence(self._h, i).decode("utf-8")

    def close(self):
        if self._h:
            bridge.lib().sw_fasta_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def search_fasta(query_path, database_path, match_score=2, mismatch_penalty=1, gap_penalty=1, top_k=5,
                 output_path="results/"):
    """What `python -m cli --query Q --database D --top-k K` leaves on disk (cli/__main__.py:93-196), computed
    by one GPU worker without the coordinator: per query a file <output_path>/<query id>.txt with its top-k
    targets.  Scores arrive as the CLI takes them (penalties positive)."""
    L = bridge.lib()
    qf, tf = Fasta(query_path), Fasta(database_path)
    db = bridge.Database(fasta=tf)
    try:
        score = bridge.AlignmentScore(int(match_score), -int(mismatch_penalty), -int(gap_penalty))
        queries = [qf.sequence(i) for i in range(len(qf))]
        k = top_k if top_k is not None else len(tf)
        n = len(queries)
        qs = (C.c_char_p * max(n, 1))(*[q.encode("utf-8") for q in queries])
        hits = (bridge.CHit * max(n * k, 1))()
        nh = (C.c_int * max(n, 1))()
        rc = L.sw_db_search_topk_batch(db._h, qs, n, score.c(), k, hits, nh)
        if rc != 0:
            raise SwHipError(f"sw_db_search_topk_batch failed ({rc}): {bridge.last_error()}")
        for q in range(n):
            first = C.cast(C.byref(hits, q * k * C.sizeof(bridge.CHit)), C.POINTER(bridge.CHit))
            names = (C.c_char_p * max(nh[q], 1))(*[tf.ids[int(hits[q * k + i].target_index)].encode("utf-8")
                                                    for i in range(nh[q])])
            rc = L.sw_write_query_results(os.fsencode(output_path), qf.ids[q].encode("utf-8"), names, first, nh[q])
            L.sw_hits_free(first, nh[q])
            if rc != 0:
                raise SwHipError(f"sw_write_query_results failed ({rc}): {bridge.last_error()}")
    finally:
        db.close()
        qf.close()
        tf.close()
