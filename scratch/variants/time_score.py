"""Kernel time of the C2 score pass (10 kb query x 100 000 x ~1 kb) for the library named by SW_HIP_LIB."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import dlsa_amd
from dlsa_amd import bridge

n = int(os.environ.get("N_TARGETS", "100000"))
flat, offs = bridge.synth(7, n, 900, 1100)
q = bridge.synth(11, 1, 10000, 10000)[0].tobytes().decode()
db = bridge.Database(flat=flat, offsets=offs)
s = dlsa_amd.AlignmentScore(2, -1, -1)
best = 1e9
for _ in range(4):
    sc = db.scores(q, s)
    st = bridge.last_stats()
    best = min(best, st["score_kernel_ms"])
print(os.path.basename(bridge.lib_path()), "score kernel ms", round(best, 3), "checksum", int(np.asarray(sc, dtype=np.int64).sum()))
