"""Worker for tests/test_cpu_gloo_topk.py: one process per "GPU" (CPU + gloo here).  Every rank owns the shard the
library's sharder gives it of ONE seeded database (lengths + a score table per query), reports its local top-k per
query, and all ranks merge the whole query batch with dlsa_amd.shard.merge_topk_batch (one all_gather)."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlsa_amd import shard  # noqa: E402


def main():
    out_path, n_targets, k, nq = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(1234)
    lens = rng.integers(0, 300, size=n_targets)
    offsets = np.zeros(n_targets + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    scores = rng.integers(0, 50, size=(nq, n_targets))  # many ties on purpose
    shard_of = shard.plan(offsets, world)
    mine = shard.local_targets(shard_of, rank)
    calls = {"n": 0}
    orig = dist.all_gather_into_tensor

    def counting(*a, **kw):
        calls["n"] += 1
        return orig(*a, **kw)
    dist.all_gather_into_tensor = counting
    local_hits = []
    for q in range(nq):
        loc = sorted(range(len(mine)), key=lambda i: (-int(scores[q, mine[i]]), int(mine[i])))[:k]
        local_hits.append([dict(score=int(scores[q, mine[i]]), target_index=i,
                                max_x=(1 << 64) - 1 if scores[q, mine[i]] == 0 else int(mine[i]), max_y=2 * int(mine[i]))
                           for i in loc])
    merged = shard.merge_topk_batch(local_hits, k, mine, dist, "cpu")
    dist.barrier()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"merged": merged, "collectives": calls["n"], "owned": [int(len(mine))]}, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
