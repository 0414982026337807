"""Sharding a target database over the GPUs of one node and merging the per-GPU top-k.

The reference distributes work by (query, target) pair over worker processes
(master/work_package/_scheduler/proportional_work_scheduler.py:30-55) and computes top-k client-side after
every pair's result has travelled worker -> master -> CLI (cli/__main__.py:153-173).  Here one process per GPU
owns a contiguous shard of the targets; the only exchange is an all_gather of k fixed-size hit records per
rank (RCCL when the tensors are on the GPUs, gloo on CPUs in the tests), merged by
(score descending, global target index ascending) -- the declared replacement of the reference's
arrival-order ties.
"""
import torch


def shard_range(n_targets, rank, world):
    """Contiguous, balanced split: rank r owns [first, first + count)."""
    base, rem = divmod(n_targets, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


RECORD_FIELDS = ("score", "target_index", "max_x", "max_y")


def merge_topk(local_hits, k, first_index=0, dist=None, device="cpu"):
    """local_hits: list of dicts with score / target_index (shard-local) / max_x / max_y, already the
    rank's own top-k.  Returns the global top-k as a list of [score, global_index, max_x, max_y]."""
    rec = torch.full((k, 4), -1, dtype=torch.int64)
    for i, h in enumerate(local_hits[:k]):
        # max_x / max_y are uint64 in the ABI (2^64-1 = no hit); stored as their int64 bit pattern
        rec[i] = torch.tensor([h["score"], first_index + h["target_index"],
                               _as_i64(h["max_x"]), _as_i64(h["max_y"])], dtype=torch.int64)
    n_valid = torch.tensor([min(len(local_hits), k)], dtype=torch.int64)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        rec, n_valid = rec.to(device), n_valid.to(device)
        world = dist.get_world_size()
        allrec = [torch.empty_like(rec) for _ in range(world)]
        allcnt = [torch.empty_like(n_valid) for _ in range(world)]
        dist.all_gather(allrec, rec)
        dist.all_gather(allcnt, n_valid)
        rows = []
        for r, c in zip(allrec, allcnt):
            rows += r[: int(c.item())].cpu().tolist()
    else:
        rows = rec[: int(n_valid.item())].tolist()
    rows.sort(key=lambda r: (-r[0], r[1]))
    return rows[:k]


def _as_i64(v):
    v = int(v)
    return v - (1 << 64) if v >= (1 << 63) else v
