"""`python -m dlsa_amd ...`: the thin drivers either side of the hot path, without the coordinator.

  benchmark            what cmd/worker/main.go:42-49 does before registering: the capacity figure and the JSON body
                       of POST /worker/register (master_rest_client.go:44-46).  --node: all GPUs of the machine;
                       --protocol: the reference's own one-pair protocol (BenchmarkRust(10 ms, 1, 2)).
  work PACKAGE.json    the loop body of cmd/worker/main.go:68-77 for ONE work package read from a file instead of
                       GET /work/: a CompleteWorkPackage (master_rest_client.go:30-42: id, job_id, queries,
                       match_score, mismatch_penalty, gap_penalty, sequences) goes through ExecuteWorkInParallel and
                       every body the worker would POST to /work/{id}/result is written as one JSON line.
  search               the CLI's flags (cli/__main__.py:97-106) minus --server-url: --query / --database FASTA files
                       in, <output-path>/<query id>.txt out, computed by this GPU worker directly.

The REST client, registration and the polling loop stay out of scope; this is the part of the two
programs that touches the path.  Nothing here computes an alignment: everything goes through libsw_hip.so.
"""
import argparse
import json
import sys

from . import bridge, worker


def _cmd_benchmark(args):
    if args.protocol:
        value = worker.benchmark_rust()
    elif args.node:
        node = bridge.Node()
        try:
            value = node.benchmark_cups()
        finally:
            node.close()
    else:
        value = worker.benchmark_capacity()
    print(f"Worker benchmark: {value / 1e6:.0f} MCUPS", file=sys.stderr)  # cmd/worker/main.go:58-60
    print(worker.machine_specs_request(value))
    return 0


def _cmd_work(args):
    with open(args.package, "r", encoding="utf-8") as f:
        d = json.load(f)
    base = worker.WorkPackage.from_json(d)
    work = worker.CompleteWorkPackage(base.id, base.job_id, base.queries, base.match_score, base.mismatch_penalty,
                                      base.gap_penalty, dict(d.get("sequences", {})))
    out = open(args.out, "w", encoding="utf-8") if args.out else sys.stdout
    node = bridge.Node() if args.node else None
    try:
        w = worker.Worker(send_result=lambda r: out.write(r.to_json() + "\n"), node=node)
        print(f"Calculating {len(work.queries)} queries", file=sys.stderr)  # cmd/worker/main.go:74
        w.execute_work_in_parallel(work, args.cpus)
        failed = getattr(w, "failed_pairs", 0)
        if failed:
            print(f"{failed} pairs failed on every entry point (entry -1, empty strings)", file=sys.stderr)
    finally:
        if node is not None:
            node.close()
        if out is not sys.stdout:
            out.close()
    return 0


def _cmd_search(args):
    worker.search_fasta(args.query, args.database, int(args.match_score), int(args.mismatch_penalty),
                        int(args.gap_penalty), args.top_k, args.output_path)
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m dlsa_amd", description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    b = sub.add_parser("benchmark", help="capacity figure a GPU worker registers with the master")
    b.add_argument("--node", action="store_true", help="all GPUs of this machine (sw_node_benchmark_cups)")
    b.add_argument("--protocol", action="store_true", help="the reference's one-pair protocol (sw_benchmark_rust)")
    b.set_defaults(fn=_cmd_benchmark)
    w = sub.add_parser("work", help="execute one CompleteWorkPackage JSON file")
    w.add_argument("package")
    w.add_argument("--out", default=None, help="file for the WorkResult bodies (default: stdout), one JSON per line")
    w.add_argument("--cpus", type=int, default=0, help="n_cpus of cmd/worker/main.go:30-40 (groups the posts only)")
    w.add_argument("--node", action="store_true", help="spread the package over all GPUs (sw_node_execute_work)")
    w.set_defaults(fn=_cmd_work)
    s = sub.add_parser("search", help="FASTA query and database in, top-k result files out")
    s.add_argument("--query", type=str, required=True, help="Path to query FASTA file")
    s.add_argument("--database", type=str, required=True, help="Path to database FASTA file")
    s.add_argument("--output-path", type=str, required=False, help="Path to output file", default="results/")
    s.add_argument("--match-score", type=str, required=False, help="Match score", default=2)
    s.add_argument("--mismatch-penalty", type=str, required=False, help="Mismatch penalty", default=1)
    s.add_argument("--gap-penalty", type=str, required=False, help="Gap penalty", default=1)
    s.add_argument("--top-k", type=int, required=False, help="Top k query matches", default=None)
    s.set_defaults(fn=_cmd_search)
    args = ap.parse_args(argv)
    try:
        return args.fn(args)
    except bridge.SwHipError as e:
        print(f"error: {e}", file=sys.stderr)
        return 1


if __name__ == "__main__":
    sys.exit(main())
