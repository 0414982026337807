"""Multi-GPU path on CPU: world_size 2 and 3 over gloo.  The database is sharded by target, every rank
keeps a local top-k, the only exchange is the all_gather of k records; the merged list must equal the top-k of
the whole table under (score desc, global index asc)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,n,k", [(2, 1001, 5), (3, 64, 7), (2, 3, 5)])
def test_sharded_topk_merge(tmp_path, world, n, k):
    out = tmp_path / "merged.json"
    port = 29500 + (os.getpid() % 500) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "_gloo_topk_worker.py"), str(out), str(n), str(k)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run(cmd, check=True, timeout=240, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    merged = json.loads(out.read_text())
    scores = np.random.default_rng(1234).integers(0, 50, size=n)
    exp = sorted(range(n), key=lambda i: (-int(scores[i]), i))[:k]
    assert [m[1] for m in merged] == exp
    assert [m[0] for m in merged] == [int(scores[i]) for i in exp]
