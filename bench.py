#!/usr/bin/env python3
"""bench.py -- GCUPS of the Smith-Waterman database search on N MI355X of one node.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): 1 query (10 kb DNA)
vs 100,000 targets (1 kb) PER GPU, match 2 / mismatch 1 / gap 1, top-k 5.  Synthetic sequences with the
distribution of the reference's utils/generate_synthetic_data.py:7-12 (i.i.d. uniform A/C/G/T), seeded.

One "step" = one full search of the rank's resident database shard through the C ABI
(sw_db_search_topk): query upload, inter-task packed-int16 score kernel over all pairs, int32 re-runs,
top-k selection, exact end-cell + traceback kernels for the 5 hits, results back on the host; with
N > 1 the ranks then exchange their k hit records (one small all_gather over RCCL) and merge them by
(score desc, global target index asc).  The database shard is already resident in HBM when the timed
region starts.  Scaling is weak: every rank owns 100,000 targets, no data-path collective.

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

Q_LEN = 10_000
T_LEN = 1_000
N_TARGETS = 100_000
TOP_K = 5
MATCH, MISMATCH, GAP = 2, -1, -1           # as the worker passes them (worker_impl.go:126-127)
OPS_PER_CELL = 6                            # algorithmic i16 ops per cell
PEAK_TOPS = 78.6                            # 256 CU x 64 lanes/clk x 2 (packed) x 2.4 GHz, BASELINE.md 3
SEED = 0xD15A
# HBM-side bytes of one score-kernel launch on the default workload, from the PMC passes committed under
# profiles/ (r01_pmc_summary.csv: WRITE_SIZE 6.32e6 KB + 2 x FETCH_SIZE 3.16e6 KB, the gfx950 half-count
# corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes); counters cannot be read from inside
# this process, so the figure is only attached when the shape is the profiled one.
TRAFFIC_DEFAULT_BYTES = 6.3164e9 + 2 * 3.1593e9
# Executed packed-i16 lane operations per cell on the default workload, from the same PMC passes:
# SQ_INSTS_VALU 3.7188e10 wave-instructions x 64 lanes x 2 halves / 1e12 cells.  The algorithmic count the
# roofline convention uses is 6 per cell; the kernel gets by with fewer (the saturating subtract
# is also the max with 0, the running maximum is taken on every second column), which is why `frac` can pass 1.
EXECUTED_OPS_PER_CELL_DEFAULT = 3.7188e10 * 128 / 1e12


def synth(rng, n, length):
    """n sequences of `length` i.i.d. uniform A/C/G/T (utils/generate_synthetic_data.py:11)."""
    lut = np.frombuffer(b"ATCG", dtype=np.uint8)
    flat = lut[rng.integers(0, 4, size=n * length, dtype=np.uint8)]
    offsets = np.arange(n + 1, dtype=np.uint64) * np.uint64(length)
    return flat, offsets


def cpu_baseline(query, flat, offsets, n_pairs, threads):
    """The oracle (CPU restatement of the reference's Rust path, driven like the Go worker) on a bounded
    sample of the same workload.  Reported baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    targets = [flat[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(n_pairs)]
    secs, cells, _ = oracle_lib.bench_pairs([query] * n_pairs, targets, GAP, MATCH, MISMATCH, threads)
    return {"value": cells / secs / 1e9, "unit": "GCUPS", "cores": threads, "kind": "port",
            "sample": f"first {n_pairs} of the {N_TARGETS} pairs (10 kb x 1 kb each), {threads} threads, "
                      f"static chunks, one find_alignment_low_memory restatement call per pair, {secs:.2f} s"}


def main():
    global Q_LEN, T_LEN
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--targets", type=int, default=N_TARGETS, help="targets per GPU (default = the named workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tlen", type=int, default=T_LEN, help="target length (default = the named workload)")
    ap.add_argument("--qlen", type=int, default=Q_LEN, help="query length (default = the named workload)")
    args = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    import dlsa_amd
    from dlsa_amd import bridge
    L = bridge.lib()
    rc = L.sw_hip_init(local_rank)
    if rc != 0:
        raise SystemExit(f"sw_hip_init({local_rank}) failed: {bridge.last_error()}")

    n_targets = args.targets
    Q_LEN, T_LEN = args.qlen, args.tlen
    # the query is the same on every rank; every rank owns its own shard of targets
    query = synth(np.random.default_rng(SEED), 1, Q_LEN)[0].tobytes()
    flat, offsets = synth(np.random.default_rng(SEED + 1 + rank), n_targets, T_LEN)
    db = bridge.Database(flat=flat, offsets=offsets)
    score = dlsa_amd.AlignmentScore(MATCH, MISMATCH, GAP)
    dev = torch.device("cpu") if rehearsal else torch.device("cuda", local_rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from dlsa_amd import shard
    first_index = rank * n_targets  # weak scaling: every rank owns n_targets targets of the global database

    def step():
        hits = db.search_topk(query, score, TOP_K)
        st = bridge.last_stats()
        merged = shard.merge_topk(hits, TOP_K, first_index, dist, dev)
        return merged, st

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    last = None
    for _ in range(args.steps):
        last, st = step()
        kernel_ms.append(st["score_kernel_ms"])
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    cells_per_rank = float(Q_LEN) * float(T_LEN) * n_targets
    total_cells = cells_per_rank * world * args.steps
    value = total_cells / elapsed / 1e9

    if rank == 0:
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else 0.0
        achieved = cells_per_rank * OPS_PER_CELL / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        default_shape = (Q_LEN, T_LEN, n_targets) == (10_000, 1_000, N_TARGETS)  # the shape the PMC passes ran on
        out = {
            "metric": "GCUPS (whole node) at 1/2/4/8 MI355X; % of packed-int16 VALU peak",
            "value": value, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i16 (packed, int32 re-run on saturation)", "data": "synthetic",
            "config": {"workload": f"1 query ({Q_LEN} nt) vs {n_targets} targets ({T_LEN} nt) per GPU, "
                                   f"match 2 / mismatch 1 / gap 1, top-k {TOP_K} (BASELINE.json configs[1])",
                       "pairs_per_gpu": n_targets, "cells_per_gpu_per_step": cells_per_rank,
                       "parallelism": f"targets sharded over {world} GPU(s), top-k gather"},
            "roofline": {"bound": "valu-int16", "achieved": achieved, "peak": PEAK_TOPS, "unit": "T i16-op/s",
                         "frac": achieved / PEAK_TOPS,
                         "traffic": TRAFFIC_DEFAULT_BYTES if default_shape else None,
                         "kernel": "sw_score_kernel", "kernel_ms": k_ms,
                         "executed": ({"ops_per_cell": EXECUTED_OPS_PER_CELL_DEFAULT,
                                       "achieved": achieved * EXECUTED_OPS_PER_CELL_DEFAULT / OPS_PER_CELL,
                                       "frac": achieved * EXECUTED_OPS_PER_CELL_DEFAULT / OPS_PER_CELL / PEAK_TOPS}
                                      if default_shape else None),
                         "note": "6 algorithmic i16 ops/cell x Q*T cells per launch / HIP-event kernel time; "
                                 "the kernel executes 4.76 i16 lane-ops per cell (`executed`), so frac can pass 1"},
            "top_hit": (dict(zip(shard.RECORD_FIELDS, last[0])) if last else None),
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = max(1, min(16, (os.cpu_count() or 2)) - 1)
            out["cpu_baseline"] = cpu_baseline(query, flat, offsets, 72 * threads, threads)
        print(json.dumps(out))
    db.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
