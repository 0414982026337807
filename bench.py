#!/usr/bin/env python3
"""bench.py -- GCUPS of the Smith-Waterman database search on N MI355X of one node.

Workloads (BASELINE.json configs; sequences from the library's seeded restatement of the reference's
utils/generate_synthetic_data.py:7-12, sw_synth_*: i.i.d. uniform A/T/C/G):
  c2 (default, the configuration the metric is quoted on): 1 query (10 kb) vs 100,000 targets (1 kb) PER GPU,
      match 2 / mismatch 1 / gap 1, top-k 5.  Weak scaling: the global database has N x 100,000 targets.
  c3: 32 queries (1 kb) vs ONE database of 1,000,000 targets of 200..5000 nt, sharded over the N GPUs.  Strong
      scaling: the total work is fixed.

One process per GPU.  Every rank derives the same global database description from the seed, asks the library's
sharder (sw_shard_plan: residue-balanced within length buckets) which targets are its own, generates and uploads
only those.  One "step" = one search of the whole query batch through the C ABI (sw_db_search_topk_batch): query
upload, inter-task packed-int16 score kernel over all pairs, int32 re-runs, on-device top-k selection, exact end
cell + traceback for the rank's hits, results back on the host; with N > 1 the ranks then exchange their
n_queries x k hit records in ONE all_gather (RCCL) and merge them by (score desc, global target index asc).  The
shard is already resident in HBM when the timed region starts.

GCUPS = sum over pairs of Q*T (unpadded, as internal/worker/worker_rust_benchmark.go:31 counts cells)
/ wall seconds / 1e9, wall = max over ranks between two barriers.

Launch: `python bench.py` (N=1) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
 bench.py --gpus N --steps K --warmup W`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TOP_K = 5
MATCH, MISMATCH, GAP = 2, -1, -1           # as the worker passes them (worker_impl.go:126-127)
PEAK_TOPS = 78.6                            # 256 CU x 64 lanes/clk x 2 (packed) x 2.4 GHz, BASELINE.md 3
SURVEY_OPS_PER_CELL = 6                     # the survey's count of algorithmic i16 ops per cell
USEFUL_OPS_PER_CELL = 4.5                   # what this formulation needs at least
SEED = 0xD15A
WORKLOADS = {
    # name: (queries, query length, targets (per GPU for weak / total for strong), min len, max len, scaling)
    "c2": dict(n_queries=1, q_len=10_000, targets=100_000, t_min=1_000, t_max=1_000, scaling="weak",
               config="BASELINE.json configs[1]"),
    "c3": dict(n_queries=32, q_len=1_000, targets=1_000_000, t_min=200, t_max=5_000, scaling="strong",
               config="BASELINE.json configs[2]"),
}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_leg(name, queries, targets, scores, threads, note):
    """The oracle (CPU restatement of the reference's Rust path, driven like the Go worker: static chunks, one
    fallback-chain call per pair) on a list of pairs.  Reported baseline only."""
    import oracle_lib
    gap, match, miss = scores
    secs, cells, _ = oracle_lib.bench_pairs(queries, targets, gap, match, miss, threads)
    return {"shape": name, "value": cells / secs / 1e9, "unit": "GCUPS", "cores": threads, "kind": "port",
            "pairs": len(queries), "seconds": round(secs, 3), "sample": note}


def cpu_baseline(bridge, main_query, main_flat, main_offsets, n_total):
    """Bounded samples of the bench workload and of the other BASELINE.json shapes on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    threads = max(1, min(16, (os.cpu_count() or 2)) - 1)
    S = (GAP, MATCH, MISMATCH)
    seqs = lambda flat, off, idx: [flat[int(off[i]):int(off[i + 1])].tobytes() for i in idx]  # noqa: E731
    n_main = 72 * threads
    main = _oracle_leg("c2", [main_query] * n_main, seqs(main_flat, main_offsets, range(n_main)), S, threads,
                       f"first {n_main} of the {n_total} pairs (10 kb x 1 kb each), {threads} threads, static chunks, "
                       f"one find_alignment_low_memory restatement call per pair")
    others = []
    # C1, full: 1 query (1 kb) vs 1000 targets of 900..1100 nt
    q, _ = bridge.synth(SEED + 10, 1, 1000, 1000)
    f, o = bridge.synth(SEED + 11, 1000, 900, 1100)
    others.append(_oracle_leg("c1", [q.tobytes()] * 1000, seqs(f, o, range(1000)), S, threads, "full: 1 x 1 kb vs 1000 x ~1 kb"))
    # C3, sub-sampled: 32 queries (1 kb) vs 10^6 targets of 200..5000 nt -> the first 12 targets for every query
    qf, qo = bridge.synth(SEED + 20, 32, 1000, 1000)
    f, o = bridge.synth(SEED + 21, 150, 200, 5000)
    qs = seqs(qf, qo, range(32))
    others.append(_oracle_leg("c3", [x for x in qs for _ in range(150)], seqs(f, o, range(150)) * 32, S, threads,
                              "sub-sampled: 32 queries x the first 150 of 10^6 targets (factor 1 / 6 667 of the pairs)"))
    # C5, sub-sampled: 150 nt queries vs 150 nt reads
    qf, qo = bridge.synth(SEED + 30, 40, 150, 150)
    f, o = bridge.synth(SEED + 31, 10000, 150, 150)
    qs = seqs(qf, qo, range(40))
    others.append(_oracle_leg("c5", [x for x in qs for _ in range(10000)], seqs(f, o, range(10000)) * 40, S, threads,
                              "sub-sampled: 40 of 1000 queries x 10 000 of 10^7 reads (factor 1 / 25 000 of the pairs)"))
    # the worker's capacity benchmark shape (worker_rust_benchmark.go:9-23): one pair, one thread, match 1 / -1 / -2
    others.append(_oracle_leg("capacity", [b"TTTTCA" + b"A" * 4096], [b"TTTTCA" + b"T" * (1 << 18)], (-2, 1, -1), 1,
                              "full: the benchmark's first pair, 4102 x 262150, one thread (ring-buffer path)"))
    main["cpu_model"] = cpu_model()
    main["others"] = others
    return main


def _sum_stats(per_shard):
    """Counters of a sharded search: sums over the shards; per-launch constants are taken from shard 0."""
    out = dict(per_shard[0])
    for st in per_shard[1:]:
        for k, v in st.items():
            if k not in ("score_packed_per_row", "score_aux_per_2rows", "score_workgroups", "pad_"):
                out[k] += v
    return out


def load_pmc_traffic(key):
    """HBM bytes per score-kernel launch from the committed PMC passes (profiles/r03_pmc.json), if the workload is
    one that was profiled; counters cannot be read from inside this process."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_pmc.json")) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--targets", type=int, default=None, help="targets per GPU (c2) / in total (c3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tlen", type=int, default=None, help="fixed target length (default: the workload's range)")
    ap.add_argument("--qlen", type=int, default=None, help="query length (default: the workload's)")
    ap.add_argument("--queries", type=int, default=None, help="number of queries (default: the workload's)")
    ap.add_argument("--node-shards", type=int, default=0,
                    help="ONE process drives S logical shards through sw_node_* (the in-process multi-GPU path; on a "
                         "one-GPU box all shards sit on device 0)")
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.workload])
    if args.targets:
        wl["targets"] = args.targets
    if args.tlen:
        wl["t_min"] = wl["t_max"] = args.tlen
    if args.qlen:
        wl["q_len"] = args.qlen
    if args.queries:
        wl["n_queries"] = args.queries

    # Libraries print to stdout (RCCL's version banner at init): keep file descriptor 1 for the ONE JSON line and
    # send everything else to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # Rehearsal of the N > 1 path on a ONE-GPU box (SW_BENCH_REHEARSAL=1): every rank uses device 0 and the
    # hit records travel over gloo, because RCCL refuses two ranks on one device.  Never used by the driver.
    rehearsal = os.environ.get("SW_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    # SW_BENCH_FORCE_DIST=1: take the N > 1 branch (process group on RCCL, sharder, device-tensor all_gather, all_reduce)
    # even with ONE rank, so that every line of it has run on the hardware a one-GPU box offers.
    force_dist = os.environ.get("SW_BENCH_FORCE_DIST") == "1"
    if force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    import dlsa_amd
    from dlsa_amd import bridge, shard
    L = bridge.lib()
    rc = L.sw_hip_init(local_rank)
    if rc != 0:
        raise SystemExit(f"sw_hip_init({local_rank}) failed: {bridge.last_error()}")

    # ---- the global database, described by (seed, n, length range); this rank's shard of it
    weak = wl["scaling"] == "weak"
    n_global = wl["targets"] * (world if weak else 1)
    q_flat, q_off = bridge.synth(SEED, wl["n_queries"], wl["q_len"], wl["q_len"])
    queries = [q_flat[int(q_off[i]):int(q_off[i + 1])].tobytes() for i in range(wl["n_queries"])]
    if wl["t_min"] == wl["t_max"]:
        lens = np.full(n_global, wl["t_min"], dtype=np.uint32)
    else:
        lens = bridge.synth_lengths(SEED + 1, n_global, wl["t_min"], wl["t_max"])
    g_off = np.zeros(n_global + 1, dtype=np.uint64)
    g_off[1:] = np.cumsum(lens, dtype=np.uint64)
    if world > 1 or force_dist:
        mine = shard.local_targets(shard.plan(g_off, world), rank)
    else:
        mine = np.arange(n_global, dtype=np.int64)
    flat, offsets = bridge.synth(SEED + 1, 0, wl["t_min"], wl["t_max"], indices=mine)
    t0 = time.perf_counter()
    node = ndb = db = None
    if args.node_shards > 0:
        node = bridge.Node(devices=[local_rank] * args.node_shards)
        ndb = bridge.NodeDatabase(node, flat=flat, offsets=offsets)
    else:
        db = bridge.Database(flat=flat, offsets=offsets)
    db_build_s = time.perf_counter() - t0
    score = dlsa_amd.AlignmentScore(MATCH, MISMATCH, GAP)
    dev = torch.device("cpu") if rehearsal else torch.device("cuda", local_rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if ndb is not None:
            hits = ndb.search_topk_batch(queries, score, TOP_K)
            st = _sum_stats([ndb.last_stats(i) for i in range(args.node_shards)])
        else:
            hits = db.search_topk_batch(queries, score, TOP_K)
            st = bridge.last_stats()
        merged = shard.merge_topk_batch(hits, TOP_K, mine, dist, dev, always=force_dist)
        return merged, st

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    stats = []
    last = None
    for _ in range(args.steps):
        last, st = step()
        stats.append(st)
    barrier()
    elapsed = time.perf_counter() - t0
    my_cells = float(wl["q_len"]) * float(offsets[-1]) * wl["n_queries"]
    if dist is not None:
        tmax = torch.tensor([elapsed, 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0].item())
        tsum = torch.tensor([my_cells], dtype=torch.float64, device=dev)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        step_cells = float(tsum.item())
    else:
        step_cells = my_cells
    value = step_cells * args.steps / elapsed / 1e9

    if rank == 0:
        # ---- roofline of the dominant kernel (sw_score_kernel), from what THIS run launched and timed:
        # HIP-event time on the library's stream and the library's own launch counters (include/sw_hip.h)
        launches = sum(s["score_launches"] for s in stats) or 1
        k_ms = sum(s["score_kernel_ms"] for s in stats) / launches
        cells_launch = sum(s["score_cells"] for s in stats) / launches
        packed, aux2 = stats[-1]["score_packed_per_row"], stats[-1]["score_aux_per_2rows"]
        wave_instr = (sum(s["score_strip_rows"] for s in stats) * (packed + aux2 / 2.0)
                      + sum(s["score_chain_rows"] for s in stats)) / launches
        k_s = k_ms * 1e-3 if k_ms > 0 else float("inf")
        executed = wave_instr * 128 / k_s / 1e12           # 64 lanes x 2 int16 halves per wave-instruction
        useful = cells_launch * USEFUL_OPS_PER_CELL / k_s / 1e12
        survey = cells_launch * SURVEY_OPS_PER_CELL / k_s / 1e12
        traffic_model = (sum(s["score_chain_rows"] for s in stats) / launches * 256 * 2
                         + float(offsets[-1]) / 4 + len(mine) * 4)
        pmc = load_pmc_traffic(f"{args.workload}:{wl['q_len']}x{len(mine)}x{wl['t_min']}-{wl['t_max']}")
        q_desc = f"{wl['n_queries']} quer{'y' if wl['n_queries'] == 1 else 'ies'} ({wl['q_len']} nt)"
        t_desc = (f"{wl['t_min']} nt" if wl["t_min"] == wl["t_max"] else f"{wl['t_min']}-{wl['t_max']} nt")
        out = {
            "metric": "GCUPS (whole node) at 1/2/4/8 MI355X; % of packed-int16 VALU peak",
            "value": value, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": wl["scaling"],
            "vs_baseline": None, "dtype": "i16 (packed, int32 re-run on saturation)", "data": "synthetic",
            "config": {"workload": f"{q_desc} vs {wl['targets']} targets ({t_desc}) "
                                   f"{'per GPU' if weak else 'in total, sharded by target'}, match 2 / mismatch 1 / gap 1, "
                                   f"top-k {TOP_K} ({wl['config']})",
                       "name": args.workload, "targets_global": n_global, "targets_rank0": int(len(mine)),
                       "cells_per_step": step_cells,
                       "parallelism": (f"one process, {args.node_shards} logical shard(s) via sw_node_*" if ndb is not None else
                                       f"one database sharded by target over {world} GPU(s) (sw_shard_plan), "
                                       f"one all_gather of {wl['n_queries']} x {TOP_K} records"
                                       + (" (SW_BENCH_FORCE_DIST: the collective path taken with one rank)" if force_dist and world == 1 else ""))},
            "backend": (dist.get_backend() if dist is not None else None),
            # value_cold: the same step including the one-time sw_db_create of the shard (validation, length sort,
            # 2-bit packing on host threads, upload); never `value`
            "value_cold": step_cells / (db_build_s + elapsed / args.steps) / 1e9, "db_build_s": db_build_s,
            "roofline": {"bound": "valu-int16", "kernel": "sw_score_kernel", "kernel_ms": k_ms,
                         "launches_per_step": launches / max(1, len(stats)),
                         # achieved / frac: packed VALU wave-instructions the launch EXECUTED in its row loop
                         # (strips x rows x instructions per row, counted by the library; instruction counts read off
                         # the generated loop) x 128 int16 lane-ops / kernel time.  Cannot exceed the issue peak.
                         "achieved": executed, "peak": PEAK_TOPS, "unit": "T i16-op/s", "frac": executed / PEAK_TOPS,
                         "useful": {"ops_per_cell": USEFUL_OPS_PER_CELL, "achieved": useful, "frac": useful / PEAK_TOPS},
                         "survey_convention": {"ops_per_cell": SURVEY_OPS_PER_CELL, "achieved": survey,
                                               "frac": survey / PEAK_TOPS,
                                               "note": "SURVEY.md 8d prices a cell at 6 ops; the kernel needs 4.5, so "
                                                       "this fraction can pass 1 and is kept for continuity only"},
                         "cells_per_launch": cells_launch, "tcups_kernel": cells_launch / k_s / 1e12,
                         "wave_instructions_per_launch": wave_instr, "workgroups": stats[-1]["score_workgroups"],
                         "traffic": (pmc or {}).get("bytes"), "traffic_source": (pmc or {}).get("source"),
                         "traffic_model": traffic_model},
            "top_hit": (dict(zip(shard.RECORD_FIELDS, last[0][0])) if last and last[0] else None),
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
            out["cpu_baseline"] = cpu_baseline(bridge, queries[0], flat, offsets, len(mine))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    for h in (db, ndb, node):
        if h is not None:
            h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
