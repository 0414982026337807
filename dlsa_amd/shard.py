"""One process per GPU (bench.py under torchrun): sharding ONE target database over the ranks with the library's
own sharder and merging the per-rank top-k of a whole query batch with a single all_gather.

The reference distributes work by (query, target) pair over worker processes
(master/work_package/_scheduler/proportional_work_scheduler.py:30-55) and computes top-k client-side after
every pair's result has travelled worker -> master -> CLI (cli/__main__.py:153-173).  Here rank r owns the targets
`sw_shard_plan` (include/sw_hip.h; residue-balanced within length buckets) assigns to shard r -- the same plan the
in-process multi-GPU handle (sw_node_*) uses -- and the only exchange is ONE all_gather of n_queries x k fixed-size
records per rank (RCCL when the tensors are on the GPUs, gloo on CPUs in the tests), merged by
(score descending, global target index ascending): the declared replacement of the reference's arrival-order ties
(SURVEY.md A10).
"""
import numpy as np
import torch

from . import bridge

RECORD_FIELDS = ("score", "target_index", "max_x", "max_y")


def plan(offsets, world):
    """shard_of[i] for every target of the global database (host arithmetic in libsw_hip.so, no GPU)."""
    return bridge.shard_plan(offsets, world)


def local_targets(shard_of, rank):
    """global indices of the targets rank `rank` owns, ascending: local index i <-> global index result[i]"""
    return np.flatnonzero(np.asarray(shard_of) == rank).astype(np.int64)


def merge_topk_batch(local_hits, k, global_index, dist=None, device="cpu", always=False):
    """local_hits[q]: this rank's own top-k for query q (dicts with score / target_index (shard-local) / max_x /
    max_y).  Returns, per query, the global top-k as [score, global_index, max_x, max_y] rows.  One collective
    (skipped for a single rank unless `always`)."""
    nq = len(local_hits)
    rec = torch.full((max(nq, 1), k, 4), -1, dtype=torch.int64)
    for q, hits in enumerate(local_hits):
        for i, h in enumerate(hits[:k]):
            # max_x / max_y are uint64 in the ABI (2^64-1 = no hit); stored as their int64 bit pattern.  A row
            # whose global index is -1 is padding (fewer than k hits).
            rec[q, i, 0] = h["score"]
            rec[q, i, 1] = int(global_index[h["target_index"]])
            rec[q, i, 2] = _as_i64(h["max_x"])
            rec[q, i, 3] = _as_i64(h["max_y"])
    if dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or always):
        rec = rec.to(device)
        world = dist.get_world_size()
        allrec = torch.empty((world * rec.shape[0], k, 4), dtype=rec.dtype, device=rec.device)  # rank-major concat
        dist.all_gather_into_tensor(allrec, rec)
        rows = allrec.cpu().numpy().reshape(world, rec.shape[0], k, 4)   # one device-to-host copy
    else:
        rows = rec.numpy()[None]
    out = []
    for q in range(nq):
        cand = rows[:, q].reshape(-1, 4)
        cand = cand[cand[:, 1] >= 0]
        order = np.lexsort((cand[:, 1], -cand[:, 0]))[:k]
        out.append(cand[order].tolist())
    return out


def _as_i64(v):
    v = int(v)
    return v - (1 << 64) if v >= (1 << 63) else v
