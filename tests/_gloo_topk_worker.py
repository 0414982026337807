"""Worker for tests/test_cpu_gloo_topk.py: one process per "GPU" (CPU + gloo here).  Every rank owns a shard
of a seeded score table, reports its local top-k, and all ranks merge with dlsa_amd.shard.merge_topk."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlsa_amd import shard  # noqa: E402


def main():
    out_path, n_targets, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    scores = np.random.default_rng(1234).integers(0, 50, size=n_targets)  # many ties on purpose
    first, count = shard.shard_range(n_targets, rank, world)
    local = sorted(range(count), key=lambda i: (-int(scores[first + i]), i))[:k]
    hits = [dict(score=int(scores[first + i]), target_index=i, max_x=(1 << 64) - 1 if scores[first + i] == 0 else i,
                 max_y=2 * i) for i in local]
    merged = shard.merge_topk(hits, k, first, dist, "cpu")
    dist.barrier()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(merged, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
